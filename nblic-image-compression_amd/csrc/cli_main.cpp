// nblic_codec_amd -- the reference tool's command line on the MI355X library (see cli.cpp).
#include "../../include/nblic_amd.h"
int main(int argc, char **argv) { return nblic_amd_cli_main(argc, argv); }
