// How finished pixels leave the streaming kernels when the caller says the frame is not read again soon (MVFX_OPT_NONTEMPORAL): WRITE-THROUGH
// with the non-temporal hint -- `global_store_dwordx4 ... sc0 sc1 nt` (round 6).
//
// Stored the ordinary way the lines of a frame sit dirty in the L2s until something evicts them or the release fence at the end of the dispatch
// walks all eight L2s to write them back; stored write-through they are in memory when the store retires, the L2s hold nothing to write back, and
// the fence finds nothing to do.  Found on the direct-dispatch lane (csrc/direct_dispatch.h), whose packets carry no release fence and therefore
// NEED such stores -- and the same stores are worth more than the missing fence in the ordinary HIP kernels: the 16 x 4K hsvfilter launch
// 87.5-87.9 k fps with `nt` stores (rounds 2-5), 86.4-86.7 k with `sc0 sc1`, 90.5 k with `sc0 sc1 nt`, 92.1-92.5 k with `sc0 sc1 nt` and CACHED
// loads, same box (profiles/r6/store_policy_ab.txt).  Not for a frame the next kernel reads: hsvfilter + hsvdetector on 16 x 1080p per launch lose
// 3 % when the filter's output is streamed past the caches (the detector then reads it from memory) -- without MVFX_OPT_NONTEMPORAL the stores
// stay ordinary cached stores.
// The inline-asm store needs the wait states the compiler inserts behind its own wide stores: a VMEM store of more than 64 bits of data reads its
// data registers up to two wait states after issue (without `s_nop 2` 0.2 % of the pixels came out wrong -- caught by the 2^24 proofs).  The
// compiler does not count an asm store in vmcnt; on gfx9 vmcnt retires in order, so its own waits only become conservative.
// (Which bit: `sc1 nt` -- agent scope -- runs like `sc0 sc1 nt`, `sc0 nt` like `nt` alone: what counts is that the store is written through
// past the L2, which on a part with eight non-coherent L2s agent scope already demands.)
// MVFX_STORE_POLICY=0 (A/B builds): `nt` stores as in rounds 2-5.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#ifndef MVFX_STORE_POLICY
#define MVFX_STORE_POLICY 1
#endif
// the LOADS of a streamed frame: 0 = ordinary cached loads (shipped: with write-through stores they beat `nt` loads, 92.1-92.5 k against 90.5 k);
// 1 = `nt` loads as in rounds 2-5 (A/B builds)
#ifndef MVFX_STREAM_NT_LOADS
#define MVFX_STREAM_NT_LOADS (MVFX_STORE_POLICY == 0)
#endif

namespace mvfx {

typedef uint32_t store_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t store_u32x3 __attribute__((ext_vector_type(3)));

// 16 bytes; STREAM: the frame is not read again soon (MVFX_OPT_NONTEMPORAL), else an ordinary cached store
template <bool STREAM>
__device__ __forceinline__ void stream_store16(void *dst, store_u32x4 t)
{
    if constexpr (!STREAM) {
        *reinterpret_cast<store_u32x4 *>(dst) = t; // (write-through WITHOUT the hint buys nothing: `sc0 sc1` alone 81 k fps on one-frame launches, as plain stores)
    } else {
#if MVFX_STORE_POLICY == 1
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 2" : : "v"(dst), "v"(t) : "memory");
#else
        __builtin_nontemporal_store(t, reinterpret_cast<store_u32x4 *>(dst));
#endif
    }
}

// 12 bytes at any 4-byte aligned address (four 3-byte pixels)
template <bool STREAM>
__device__ __forceinline__ void stream_store12(void *dst, store_u32x3 t)
{
    if constexpr (!STREAM) {
        *reinterpret_cast<store_u32x3 *>(dst) = t;
    } else {
#if MVFX_STORE_POLICY == 1
        asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1 nt\n\ts_nop 2" : : "v"(dst), "v"(t) : "memory");
#else
        __builtin_nontemporal_store(t, reinterpret_cast<store_u32x3 *>(dst));
#endif
    }
}

} // namespace mvfx
