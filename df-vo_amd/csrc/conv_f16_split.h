// f16 hi / lo plane split of fp32 operands (see conv_win_f16s.h for the arithmetic) -- shared by the f16x3 kernels.
#pragma once
// (included inside namespace dfvo)

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float F16S_LO_SCALE = 2048.0f;          // 2^11
constexpr float F16S_LO_UNSCALE = 1.0f / 2048.0f;
constexpr float F16S_MAX = 65504.0f;

__device__ __forceinline__ void split_f16_planes(f32x4 x, h16x4* hi, h16x4* lo, float& amax) {
    // (spelled as instructions: fmaxf() drags a canonicalising v_max per operand along, 7 instructions instead of 2)
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(x[0]), "v"(x[1]));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(x[2]), "v"(x[3]));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = x[e];
        const _Float16 h = (_Float16)v;  // round to nearest even
        (*hi)[e] = h;
        (*lo)[e] = (_Float16)((v - (float)h) * F16S_LO_SCALE);  // v - h is exact in fp32
    }
}

// "f16" mode (DFVO_CONV_PRECISION=f16: one product per term, BASELINE config 5's "fp16 flow"): the hi plane alone
__device__ __forceinline__ void split_f16_hi(f32x4 x, h16x4* hi, float& amax) {
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(x[0]), "v"(x[1]));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(x[2]), "v"(x[3]));
#pragma unroll
    for (int e = 0; e < 4; ++e) (*hi)[e] = (_Float16)x[e];
}

