// Helpers shared by all kernels: magic division, wave-level LDS fence, single-lane-op IEEE arithmetic, the accumulate / divide epilogue.
//
// Device code of libmemb_hip.so (gfx950 / CDNA4). Included by memb_hip.hip only,
// inside its anonymous namespace; see that file for the overview.
#pragma once

// q / d with a host-computed magic = ceil(2^32 / d) (exact while q * d < 2^32);
// magic == 0 means "no magic" (d == 1, or the range is too large): plain division.
__device__ __forceinline__ uint32_t fastDivide(uint32_t q, uint32_t magic, uint32_t d)
{
    return magic ? __umulhi(q, magic) : q / d;
}

// Orders this wave's LDS writes before its later LDS reads (and vice versa).
// LDS operations of one wave execute in order; the fence makes the compiler
// wait for them and keeps it from moving accesses across.
__device__ __forceinline__ void waveLdsFence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Single-lane-op IEEE fp32 add / sub / mul. Written as instructions because the
// optimiser otherwise pairs neighbouring operations into v_pk_add_f32 /
// v_pk_mul_f32, and the packed forms flush subnormal values on gfx950 (measured:
// min = 1e-40 came back as 0), which would break bit parity with the CPU.
__device__ __forceinline__ float addRn(float a, float b)
{
    float r;
    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float subRn(float a, float b)
{
    float r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float mulRn(float a, float b)
{
    float r;
    asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Epilogue of ReadersUnion 'average' (reference python/memb/readers_union.py:18:
// numpy.mean over the readers = fp32 sums in reader order, then one division
// by the reader count): later readers add to what earlier ones stored, the last
// one divides. Same operations and order as numpy, so the result is bit-identical.
__device__ __forceinline__ float epilogue(float value, const float* destination, uint32_t accumulate, float divisor)
{
    if (accumulate) {
        value = addRn(*destination, value);
    }
    if (divisor != 0.f) {
        value = __fdiv_rn(value, divisor);
    }
    return value;
}

__device__ __forceinline__ float4 epilogue4(float4 value, const float* destination, uint32_t accumulate, float divisor)
{
    if (accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(destination);
        value.x = addRn(old.x, value.x);
        value.y = addRn(old.y, value.y);
        value.z = addRn(old.z, value.z);
        value.w = addRn(old.w, value.w);
    }
    if (divisor != 0.f) {
        value.x = __fdiv_rn(value.x, divisor);
        value.y = __fdiv_rn(value.y, divisor);
        value.z = __fdiv_rn(value.z, divisor);
        value.w = __fdiv_rn(value.w, divisor);
    }
    return value;
}
