// ABI version + host utilities of libpfpn_hip.so.
#include "pfpn_common.h"

#include <cstddef>
#include <cstdint>

extern "C" int pfpn_abi_version(void) { return 1; }

// CRC-32C (Castagnoli, reflected 0x82F63B78), slicing-by-8 on the host: the checksum TensorFlow's
// tensor_bundle stores per tensor and per table block (reference checkpoints: models/distributed_model.py:436-496
// saves through tf.train.Saver).  `crc` chains calls: pass 0 first, then the previous return value.
namespace {
struct Crc32cTable {
    uint32_t t[8][256];
    Crc32cTable() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
    }
};
}  // namespace

extern "C" uint32_t pfpn_crc32c(const void* data, size_t n, uint32_t crc) {
    static const Crc32cTable T;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = crc ^ 0xffffffffu;
    while (n >= 8) {
        const uint32_t lo = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        const uint32_t hi = (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
        const uint32_t a = c ^ lo;
        c = T.t[7][a & 0xff] ^ T.t[6][(a >> 8) & 0xff] ^ T.t[5][(a >> 16) & 0xff] ^ T.t[4][a >> 24] ^
            T.t[3][hi & 0xff] ^ T.t[2][(hi >> 8) & 0xff] ^ T.t[1][(hi >> 16) & 0xff] ^ T.t[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}
