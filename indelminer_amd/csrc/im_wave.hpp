// im_wave.hpp -- wave-level helpers of the realignment kernels (gfx950, 64 lanes per wavefront).
#pragma once

#include "im_device.hpp"

namespace im {
namespace {

constexpr int kScoreMatch = 1;              // src/localalign.c:10-13
constexpr int kScoreMismatch = -10;

// ---- wave helpers (64 lanes) ------------------------------------------------
//
// Scans and reductions run on the DPP cross-lane path (row_shr / row_bcast), six
// VALU steps and no LDS round trip, instead of ds_bpermute shuffles.  Reductions
// end in v_readlane, so their results live in SGPRs and the control flow that
// hangs off them is scalar.

constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppWaveShr1 = 0x138, kDppWaveShl1 = 0x130;

// A workgroup is one wavefront, so no s_barrier is ever needed: LDS executes one wave's
// instructions in order, a later read sees an earlier write of any lane.  What is needed is
// only that the compiler keeps the order -- and, unlike __syncthreads(), does NOT drain the
// vector-memory queue (s_waitcnt vmcnt(0)), which is what lets reference-window loads issued
// before a phase overlap that phase.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Wave-uniform loads of read-only inputs through the scalar cache (s_load): the per-read scalars and the
// contig table are a handful of dwords that neighbouring workgroups share cache lines of, and the scalar
// path returns them in a fraction of a vector load's latency -- they head the read's dependent chain.
#define IM_CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ T sload(const T* p)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return *(const IM_CONST_AS T*)p;
#pragma clang diagnostic pop
}
__device__ __forceinline__ int64_t uni64(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_mov(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ int wave_scan_add(int v, int /*lane*/)
{
    v += dpp_mov<kDppRowShr1>(0, v);
    v += dpp_mov<kDppRowShr2>(0, v);
    v += dpp_mov<kDppRowShr4>(0, v);
    v += dpp_mov<kDppRowShr8>(0, v);
    v += dpp_mov<kDppBcast15, 0xa>(0, v);
    v += dpp_mov<kDppBcast31, 0xc>(0, v);
    return v;                                // inclusive
}
__device__ __forceinline__ int wave_scan_min_incl(int v)
{
    v = min(v, dpp_mov<kDppRowShr1>(INT_MAX, v));
    v = min(v, dpp_mov<kDppRowShr2>(INT_MAX, v));
    v = min(v, dpp_mov<kDppRowShr4>(INT_MAX, v));
    v = min(v, dpp_mov<kDppRowShr8>(INT_MAX, v));
    v = min(v, dpp_mov<kDppBcast15, 0xa>(INT_MAX, v));
    v = min(v, dpp_mov<kDppBcast31, 0xc>(INT_MAX, v));
    return v;
}
__device__ __forceinline__ int wave_scan_max_incl(int v)
{
    v = max(v, dpp_mov<kDppRowShr1>(INT_MIN, v));
    v = max(v, dpp_mov<kDppRowShr2>(INT_MIN, v));
    v = max(v, dpp_mov<kDppRowShr4>(INT_MIN, v));
    v = max(v, dpp_mov<kDppRowShr8>(INT_MIN, v));
    v = max(v, dpp_mov<kDppBcast15, 0xa>(INT_MIN, v));
    v = max(v, dpp_mov<kDppBcast31, 0xc>(INT_MIN, v));
    return v;
}
__device__ __forceinline__ int wave_scan_min_excl(int v, int /*lane*/)
{
    // exclusive running minimum, identity INT_MAX
    return dpp_mov<kDppWaveShr1>(INT_MAX, wave_scan_min_incl(v));
}
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_scan_max_incl(v), 63); }
__device__ __forceinline__ int wave_min(int v) { return __builtin_amdgcn_readlane(wave_scan_min_incl(v), 63); }
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_scan_add(v, 0), 63); }

// base2bits, src/alignment.c:11-24
__device__ __forceinline__ uint32_t code2(uint32_t c)
{
    const uint32_t u = (c | 0x20u) - 'a';
    uint32_t x = (c >> 1) & 3u;
    x ^= x >> 1;
    const bool valid = (u < 26u) && ((0x80045u >> u) & 1u);
    return valid ? x : 0u;
}

// base2bits of four ASCII bytes at once: byte j's code in bits 2j, 2j+1 of the result.  Per byte the code is
// (c >> 1) & 3 with its two upper values swapped; a byte counts only when its lower-case form IS the letter
// that code stands for (one v_perm looks the four letters up), anything else codes 0 like the reference's default.
__device__ __forceinline__ uint32_t code2x4(uint32_t v)
{
    uint32_t x = (v >> 1) & 0x03030303u;
    x ^= (x >> 1) & 0x01010101u;
    const uint32_t want = __builtin_amdgcn_perm(0u, 0x74676361u, x);           // 'a' 'c' 'g' 't' by code
    const uint32_t d = (v | 0x20202020u) ^ want;                                // zero byte: a valid letter
    const uint32_t nz = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;  // 0x80 in every non-zero byte
    x &= ~((nz >> 7) * 3u);
    return (x | (x >> 6) | (x >> 12) | (x >> 18)) & 0xFFu;
}

typedef unsigned short im_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)     // v_pk_max_u16
{
    im_u16x2 x, y;
    __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
    const im_u16x2 r = __builtin_elementwise_max(x, y);
    uint32_t o; __builtin_memcpy(&o, &r, 4);
    return o;
}

__device__ __forceinline__ uint32_t lds_byte(const uint32_t* base, uint32_t i)
{
    return reinterpret_cast<const uint8_t*>(base)[i];
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// ---- what every realignment kernel leaves in the result record ----------------

__device__ __forceinline__ void finish(im_read_result* out, int status, int n_band, int lane)
{
    if (lane == 0) { out->status = status; out->n_band = n_band; if (status != IM_ST_EVIDENCE) { out->n_ev = 0; out->n_ops = 0; out->ref_start = 0; } }
}

// evidence slots of read c (see im_dev_batch): slot k < n_ev live, the rest empty
__device__ __forceinline__ void write_slots(const RealignArgs& A, int c, int n_ev, int cls0, int b1, int b2, int lane)
{
    if (A.batch.ev_cls && lane < IM_MAX_EV) {
        const int64_t base = (int64_t)c * IM_MAX_EV;            // wave-uniform: the stores take it as their scalar base
        const bool live = lane < n_ev;
        (A.batch.ev_cls + base)[lane] = live ? cls0 : -1;
        (A.batch.ev_b1 + base)[lane] = live ? b1 : 0;
        (A.batch.ev_b2 + base)[lane] = live ? b2 : 0;
    }
}

}  // namespace
}  // namespace im
