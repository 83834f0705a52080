// bb_pack_abi.cpp — bb_pack_bases of the C ABI (include/barbell_amd.h): host code of libbarbell_amd.so, compiled by the host compiler.
#include "../../include/barbell_amd.h"
#include "bb_pack.h"

extern "C" uint64_t bb_pack_bases(const uint8_t* bases, uint64_t n, uint8_t* out) {
    if (!n) return 0;
    if (!bases || !out) return 0;
    bool zero_pair = false;   // (a pair of two non-IUPAC characters is a byte like any other here: no text framing to protect)
    return (uint64_t)pack_bases(out, bases, bases + n, 15, zero_pair);
}
