// Range slots of the fp16x2 form (kernels.h ConvParams::xr / yr): how a convolution learns the largest |pixel| of the
// tensor it reads IN THE RUN THAT USES IT, without a pass over that tensor.
//
// A slot is one word the READER takes (word 0) and RANGE_PART partial words the WRITERS raise.
// Writers: every kernel that stores a tensor some fp16x2-form convolution reads computes the largest bit pattern of |value| over
// what it stored (non-negative floats order like their bit patterns; a NaN sorts above infinity and stays visible), reduces it over the
// block (or the wavefront) and raises ONE partial word -- picked by its block index -- with an atomicMax.  Same-address atomics
// serialise at 11 ns each on MI355X (scripts/microbench/amax_ubench.hip: 262 144 of them cost 2.9 ms), so the 16 384 blocks of a res2
// layer must not meet on one word; spread over 1024 words on different lines they cost nothing.  The maximum is order-independent:
// the slot's content is a function of the tensor alone.
// Fold: a one-block kernel (misc.hip range_fold_kernel) in front of the first reader after a write takes the maximum of the partial
// words into word 0.
// Reader: ONE scalar load, then the power of two that puts the largest pixel into [2^13, 2^14): 4x of headroom to the largest half
// (the Winograd kernels spend it on their input transform's growth), full relative precision (two half terms, 22-23 bits) for every
// pixel down to 2^-17 of the largest.  (Measured on the way here, same-box A/B of the headline, scripts/ab_round.sh: readers that took
// the maximum over 32 spread sub-slots themselves cost 4.3 ms of a 62 ms step as scalar loads -- 32 serialised scalar-cache misses
// per CU and launch -- and 20 % on the short-K layers as one vector load per lane; writers whose bookkeeping was woven into the store
// loop cost 30-40 registers per lane and a resident block.)
// The plan zeroes its slots at the start of every run (a kernel of its own: a hipMemsetAsync node of a captured graph filled parts
// of the table with a stale 16-byte pattern on some replays).
#pragma once
#include <hip/hip_runtime.h>

#define RANGE_PART 1024                          // partial words of a slot
#define RANGE_PART_OFF 64                        // ... starting 256 bytes behind word 0
#define RANGE_WORDS (RANGE_PART_OFF + RANGE_PART)   // words of one slot

__device__ __forceinline__ unsigned range_abs_bits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7FFFFFFFu; }

// maximum over the wavefront (wave-uniform result; every lane of the wavefront must be active).
// Six data-parallel-primitive moves on the vector ALU (quad permutes, row mirrors, the two row broadcasts of gfx9): the same reduction
// through ds_bpermute (__shfl_xor) is a chain of six dependent LDS-pipeline round trips at the very end of a wavefront's life -- part
// of the 3.2 ms per 65 ms step the first form of the conv epilogue's note pass cost (same-box A/B, scripts/ab_round.sh).
__device__ __forceinline__ unsigned range_wave_max(unsigned u)
{
#define RANGE_DPP_MAX(ctrl, rmask)                                                                              \
    do {                                                                                                        \
        const unsigned t_ = (unsigned)__builtin_amdgcn_update_dpp((int)u, (int)u, ctrl, rmask, 0xF, false);     \
        u = t_ > u ? t_ : u;                                                                                    \
    } while (0)
    RANGE_DPP_MAX(0xB1, 0xF);       // quad_perm [1,0,3,2]
    RANGE_DPP_MAX(0x4E, 0xF);       // quad_perm [2,3,0,1]
    RANGE_DPP_MAX(0x141, 0xF);      // row_half_mirror
    RANGE_DPP_MAX(0x140, 0xF);      // row_mirror: every lane of a row of 16 holds the row's maximum
    RANGE_DPP_MAX(0x142, 0xA);      // row_bcast15 into rows 1 and 3
    RANGE_DPP_MAX(0x143, 0xC);      // row_bcast31 into rows 2 and 3: lane 63 holds the maximum of the wavefront
#undef RANGE_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)u, 63);
}

// one atomic per wavefront; `key`: any index that differs between the wavefronts of a launch (spreads them over the partial words)
__device__ __forceinline__ void range_note_wave(unsigned* slot, unsigned m, unsigned key)
{
#ifdef RANGE_AB_NO_NOTE
    return;
#endif
    m = range_wave_max(m);
#ifdef RANGE_AB_NO_ATOMIC      // timing experiment: everything but the atomic
    asm volatile("" :: "s"(m));
    return;
#endif
    if ((threadIdx.x & 63) == 0 && m) atomicMax(slot + RANGE_PART_OFF + (key & (RANGE_PART - 1)), m);
}

// one atomic per block (byte movers with hundreds of thousands of wavefronts per launch).  Every thread of the block must call it.
__device__ __forceinline__ void range_note_block(unsigned* slot, unsigned m, unsigned key)
{
#ifdef RANGE_AB_NO_NOTE
    return;
#endif
    __shared__ unsigned range_sm[16];
    m = range_wave_max(m);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) range_sm[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) m = range_sm[w] > m ? range_sm[w] : m;
        if (m) atomicMax(slot + RANGE_PART_OFF + (key & (RANGE_PART - 1)), m);
    }
    __syncthreads();      // a second call may follow: range_sm is read by thread 0 above
}

// the slot's value (after the fold): one scalar load (the address is uniform and the kernel has written nothing yet).  Same-box A/B of
// the headline (scripts/ab_round.sh): as an agent-scope vector load of the same word 71 ms per step against 67.
__device__ __forceinline__ unsigned range_read(const unsigned* __restrict__ slot)
{
#ifdef RANGE_AB_NO_READ      // timing experiments only (scripts/ab_round.sh): WRONG results
    return 0x42000000u;
#endif
    return (unsigned)__builtin_amdgcn_readfirstlane((int)slot[0]);
}

// largest |x| (bit pattern) -> s = 2^e with s * largest in [2^13, 2^14) and 1 / s; an all-zero tensor: 1
struct RangeScale { float s, inv; };
__host__ __device__ __forceinline__ RangeScale range_scale(unsigned bits)
{
#ifndef RANGE_TOP_EXP
#define RANGE_TOP_EXP 14      // the largest pixel lands in [2^(RANGE_TOP_EXP - 1), 2^RANGE_TOP_EXP)
#endif
    int e = (126 + RANGE_TOP_EXP) - (int)((bits >> 23) & 0xFFu);      // largest = m 2^(E - 126), m in [0.5, 1)
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    if (!bits) e = 0;
    RangeScale r;
    const unsigned sb = (unsigned)(127 + e) << 23, ib = (unsigned)(127 - e) << 23;
    r.s = __builtin_bit_cast(float, sb);
    r.inv = __builtin_bit_cast(float, ib);
    return r;
}

// prologue of an fp16x2-form convolution: the scale pair of its input slot (null: 1).  (A non-finite range is reported by the fold
// kernel, not here: the branch with its system-scope atomic in the prologue of the hot kernels cost 2.2 ms of a 63 ms step.)
__device__ __forceinline__ RangeScale range_prologue(const unsigned* xr)
{
    RangeScale one; one.s = 1.f; one.inv = 1.f;
    if (!xr) return one;
    return range_scale(range_read(xr));
}
