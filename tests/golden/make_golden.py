"""Regenerates tests/golden/ from the COMPILED, UNMODIFIED reference (oracle/_ref).

Runs only in the build container (needs /root/reference to have been compiled by
`make -C oracle ref`).  Emits
  small_streams.npz   reference .nblic bytes for every small case (inputs.SMALL_SHAPES x
                      CONTENTS x PARAM_CLASSES), plus QNBLIC (effort 0) streams;
  manifest.json       length + sha256 of stream and reconstruction for every case, and for the
                      larger frames (512^2 .. 4096^2, Kodak when readable) hashes only.
Contains no reference code: it calls the reference's public C API through ctypes.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import inputs  # noqa: E402
from oracle.oracle import Reference, syn1  # noqa: E402


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    ref = Reference()
    streams, manifest = {}, {"small": {}, "large": {}, "kodak_e1": {}, "q_small": {}}
    for (h, w) in inputs.SMALL_SHAPES:
        for content in inputs.CONTENTS:
            img = inputs.make(content, h, w)
            for near, effort in inputs.PARAM_CLASSES:
                cid = inputs.case_id(content, h, w, near, effort)
                s, rec, n_out, e_out = ref.encode(img, near, effort)
                streams[cid] = np.frombuffer(s, np.uint8)
                manifest["small"][cid] = {"len": len(s), "sha256": sha(s), "recon_sha256": sha(rec.tobytes()),
                                          "near_out": n_out, "effort_out": e_out}
            q = ref.qencode(img)
            qid = f"q_{content}_{h}x{w}"
            streams[qid] = np.frombuffer(q, np.uint8)
            manifest["q_small"][qid] = {"len": len(q), "sha256": sha(q)}
    large = [(512, 512, 0, 1), (512, 512, 2, 1), (1024, 1024, 0, 1), (2048, 2048, 0, 1), (4096, 4096, 0, 1),
             (256, 256, 0, 2), (256, 256, 0, 3), (256, 256, 2, 2), (768, 512, 0, 1)]
    for (h, w, near, effort) in large:
        for seed in ([1] if h * w > 1 << 20 else [1, 2, 3]):
            img = syn1(h, w, seed)
            s, rec, _, _ = ref.encode(img, near, effort)
            manifest["large"][f"syn1s{seed}_{h}x{w}_n{near}_e{effort}"] = {
                "len": len(s), "sha256": sha(s), "recon_sha256": sha(rec.tobytes()), "input_sha256": sha(img.tobytes())}
    for (h, w) in [(512, 512), (4096, 4096)]:
        q = ref.qencode(syn1(h, w, 1))
        manifest["large"][f"syn1s1_{h}x{w}_q0"] = {"len": len(q), "sha256": sha(q)}
    # Kodak (BASELINE config 3): lengths + hashes only, from images read in place (never copied)
    if os.path.isdir(inputs.KODAK_DIR):
        for name in sorted(os.listdir(inputs.KODAK_DIR)):
            img = inputs.read_gray_bmp(os.path.join(inputs.KODAK_DIR, name))
            s, _, _, _ = ref.encode(img, 0, 1)
            q = ref.qencode(img)
            manifest["kodak_e1"][name] = {"shape": list(img.shape), "input_sha256": sha(img.tobytes()), "len": len(s), "sha256": sha(s),
                                          "q_len": len(q), "q_sha256": sha(q)}
    np.savez_compressed(os.path.join(HERE, "small_streams.npz"), **streams)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(len(streams), "small streams;", len(manifest["large"]), "large hashes")


if __name__ == "__main__":
    main()
