// Thin extern "C" entry around the reference's own a-law / mu-law expander (src/io/amulaw.h:20-53).
// Compiled only by oracle/Makefile into oracle/_ref/ when /root/reference exists; used by
// tests/test_cli.py to pin the CLI's decoders byte for byte (all 256 codes, both laws).
#include "amulaw.h"

extern "C" void ref_amulaw_table(int mode /* 0 = mu-law, 1 = A-law */, short *out256) {
    for (int v = 0; v < 256; v++) {
        char a = (char)v;
        short int s = 0;
        alaw2lin(a, s, mode);
        out256[v] = s;
    }
}
