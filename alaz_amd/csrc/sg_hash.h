// sg_hash.h — hash functions shared by the kernels, the host engine and the host-side unit tests (no HIP dependency).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define SG_HD __host__ __device__ __forceinline__
#else
#define SG_HD static inline
#endif

SG_HD uint32_t sg_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
// cuckoo join table: bucket (two 8-byte entries) of an IP under the two hash functions
SG_HD uint32_t ip_h1(uint32_t ip, uint32_t bmask) { return sg_fmix32(ip) & bmask; }
SG_HD uint32_t ip_h2(uint32_t ip, uint32_t bmask) { return sg_fmix32(ip ^ 0x7F4A7C15u) & bmask; }

// block table level 1: slot of block number b = ip >> 8 (b < 2^24, so b * K is the low word of a 24 x 24 bit product:
// one full-rate v_mul_u32_u24 on the device, plain 32-bit wrap-around on the host — the same bits)
#define SG_JL1_EMPTY  0xFFFFFFFFu
#define SG_JL1_K1     0x9E3779u
#define SG_JL1_K2     0xC2B2AFu
SG_HD uint32_t jl1_h1(uint32_t b, uint32_t mask) { return ((b * SG_JL1_K1) >> 9) & mask; }
SG_HD uint32_t jl1_h2(uint32_t b, uint32_t mask) { return ((b * SG_JL1_K2) >> 11) & mask; }
