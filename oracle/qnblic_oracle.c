/* placeholder until the QNBLIC (effort 0) restatement lands */
