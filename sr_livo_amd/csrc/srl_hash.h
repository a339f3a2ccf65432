// srl_hash.h -- voxel key packing + hash shared by host (table build) and device (probe).
// Replaces std::hash<voxel> (cloudMap.h:173-184) + tsl::robin_map's bucket_for_hash
// (robin_growth_policy.h:107-109): only find() semantics are observable on the hot path, so the
// device table is free to use its own (better mixed) hash.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#define SRL_HD __host__ __device__
#else
#define SRL_HD
#endif

// voxel (cloudMap.h:124-145): three int16, packed into the low 48 bits
SRL_HD inline unsigned long long srl_pack_key(short x, short y, short z) {
    return (unsigned long long)(unsigned short)x | ((unsigned long long)(unsigned short)y << 16) |
           ((unsigned long long)(unsigned short)z << 32);
}
SRL_HD inline void srl_unpack_key(unsigned long long k, short *x, short *y, short *z) {
    *x = (short)(unsigned short)(k & 0xFFFFu);
    *y = (short)(unsigned short)((k >> 16) & 0xFFFFu);
    *z = (short)(unsigned short)((k >> 32) & 0xFFFFu);
}
// Multiplicative mix of the three int16 coordinates.  The per-coordinate products use 24-bit constants so
// the device can take the full-rate 24-bit multiplier (v_mul_u32_u24); one 32-bit multiply finalises.
SRL_HD inline unsigned srl_hash_key(unsigned long long k) {
    const unsigned x = (unsigned)(k & 0xFFFFu), y = (unsigned)((k >> 16) & 0xFFFFu), z = (unsigned)((k >> 32) & 0xFFFFu);
#ifdef __HIP_DEVICE_COMPILE__
    unsigned h = __umul24(x, 0x9E3779u) ^ __umul24(y, 0x85EBCBu) ^ __umul24(z, 0xC2B2AFu);
#else
    unsigned h = (x * 0x9E3779u) ^ (y * 0x85EBCBu) ^ (z * 0xC2B2AFu);   // 16-bit x 24-bit: no overflow, same value
#endif
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    return h;
}
