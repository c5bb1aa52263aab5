// Range slots of the fp16x2 form (kernels.h ConvParams::xr / yr): how a convolution learns the largest |pixel| of the
// tensor it reads IN THE RUN THAT USES IT, without a pass over that tensor.
//
// Writers: every kernel that stores a tensor some fp16x2-form convolution reads keeps the running maximum of the bit patterns of
// |value| over what it stored (non-negative floats order like their bit patterns; a NaN sorts above infinity and stays visible),
// reduces it over the wavefront (or the block) and raises the tensor's slot with ONE atomicMax.  Same-address atomics serialise at
// 11 ns each on MI355X (scripts/microbench/amax_ubench.hip: 262 144 of them cost 2.9 ms), so a slot is RANGE_SUB sub-slots 256 bytes
// apart -- different memory channels: 65 536 atomics per launch are free, 262 144 cost 75 us -- and a writer picks one by its
// block / wavefront index.  The maximum is order-independent: the slot's final content is a function of the tensor alone.
// Reader: the maximum over the sub-slots (one load per lane, five cross-lane steps), then the power of two that puts it into
// [2^13, 2^14): 4x of headroom to the largest half (the Winograd kernels spend it on their input transform's growth), full relative
// precision (two half terms, 22-23 bits) for every pixel down to 2^-17 of the largest.
// The plan zeroes its slots at the start of every run (one memset node in the captured graph).
#pragma once
#include <hip/hip_runtime.h>

#define RANGE_SUB 32
#define RANGE_STRIDE 64                          // words between sub-slots (256 bytes)
#define RANGE_WORDS (RANGE_SUB * RANGE_STRIDE)   // words of one slot

__device__ __forceinline__ unsigned range_abs_bits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7FFFFFFFu; }

__device__ __forceinline__ unsigned range_wave_max(unsigned u)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)u, o);
        u = t > u ? t : u;
    }
    return u;
}

// one atomic per wavefront; `key`: any index that differs between the wavefronts of a launch (spreads them over the sub-slots)
__device__ __forceinline__ void range_note_wave(unsigned* slot, unsigned m, unsigned key)
{
    m = range_wave_max(m);
    if ((threadIdx.x & 63) == 0 && m) atomicMax(slot + (key & (RANGE_SUB - 1)) * RANGE_STRIDE, m);
}

// one atomic per block (byte movers with hundreds of thousands of wavefronts per launch).  Every thread of the block must call it.
__device__ __forceinline__ void range_note_block(unsigned* slot, unsigned m, unsigned key)
{
    __shared__ unsigned range_sm[16];
    m = range_wave_max(m);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) range_sm[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) m = range_sm[w] > m ? range_sm[w] : m;
        if (m) atomicMax(slot + (key & (RANGE_SUB - 1)) * RANGE_STRIDE, m);
    }
    __syncthreads();      // a second call may follow: range_sm is read by thread 0 above
}

// the slot's value: wave-uniform (every lane of the wavefront must call it)
__device__ __forceinline__ unsigned range_read(const unsigned* slot)
{
    unsigned u = slot[(threadIdx.x & (RANGE_SUB - 1)) * RANGE_STRIDE];
#pragma unroll
    for (int o = RANGE_SUB / 2; o; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)u, o);
        u = t > u ? t : u;
    }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)u);
}

// largest |x| (bit pattern) -> s = 2^e with s * largest in [2^13, 2^14) and 1 / s; an all-zero tensor: 1
struct RangeScale { float s, inv; };
__host__ __device__ __forceinline__ RangeScale range_scale(unsigned bits)
{
    int e = 140 - (int)((bits >> 23) & 0xFFu);      // largest = m 2^(E - 126), m in [0.5, 1)
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    if (!bits) e = 0;
    RangeScale r;
    const unsigned sb = (unsigned)(127 + e) << 23, ib = (unsigned)(127 - e) << 23;
    r.s = __builtin_bit_cast(float, sb);
    r.inv = __builtin_bit_cast(float, ib);
    return r;
}

// prologue of an fp16x2-form convolution: the scale pair of its input slot (null: 1); a non-finite range is reported once per
// launch through the host-mapped flag
__device__ __forceinline__ RangeScale range_prologue(const unsigned* xr, unsigned* rflag, int op_index)
{
    RangeScale one; one.s = 1.f; one.inv = 1.f;
    if (!xr) return one;
    const unsigned bits = range_read(xr);
    if (bits >= 0x7F800000u && rflag && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        if (atomicCAS(rflag, 0u, (unsigned)op_index + 1u) == 0u) rflag[1] = bits;      // first offender: its index and what it saw
        __threadfence_system();
    }
    return range_scale(bits);
}
