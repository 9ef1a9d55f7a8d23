// Epilogue helpers shared by every convolution kernel of the library (conv_igemm_f32.hip, conv_taps_f16s.hip, conv_gemm_f32g.hip):
// bias + residual + activation + store of accumulator quads.  See conv_igemm_f32.hip for the accumulator layout.
#pragma once
// (included inside namespace dfvo)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float apply_act(float v, int act, float a) {
    switch (act) {
        case ACT_LEAKY: return v > 0.f ? v : v * a;
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_ELU: return v > 0.f ? v : a * expm1f(v);
        case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// Epilogue of four consecutive output channels [col0, col0+4) of one pixel (linear index m).  The MFMAs are issued
// with the weight fragment as the A operand, so a lane's four accumulator registers are four consecutive couts of
// ONE pixel: bias / residual / store move as 16-byte vectors (a quarter of the store instructions, whole 64-byte
// runs per pixel) whenever the layer's strides allow it.
__device__ __forceinline__ void conv_epilogue_quad(const ConvParams& p, size_t m, int col0, f32x4 a, bool vec_ok) {
    float* d = p.dst + m * p.dst_cs + p.dst_co + col0;
    if (vec_ok && col0 + 3 < p.cout) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + col0);
        f32x4 v = a + b;
        if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + m * p.res_cs + p.res_co + col0);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = apply_act(v[q], p.act, p.act_param);
        *reinterpret_cast<f32x4*>(d) = v;
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = col0 + q;
        if (col < p.cout) {
            float v = a[q] + p.bias[col];
            if (p.res) v += p.res[m * p.res_cs + p.res_co + col];
            d[q] = apply_act(v, p.act, p.act_param);
        } else if (col < p.dst_zero_to) {
            d[q] = 0.f;
        }
    }
}
__device__ __forceinline__ bool conv_vec_ok(const ConvParams& p) {
    return (((p.dst_cs | p.dst_co) & 3) == 0) && (!p.res || (((p.res_cs | p.res_co) & 3) == 0));
}


// ---- batched epilogue of the f16x3 kernels --------------------------------------------------------------------------
// conv_epilogue_quad() interleaves, per quad, a bias (and residual) LOAD with the quad's STORE.  Stores count on vmcnt on
// this architecture, so the wait for each bias value drains every store issued before it: 16-24 quads per wave become
// 16-24 serialised store round trips (measured on the level-2 layers: 40 us of a 213 us launch at two workgroups per CU,
// 111 us of 283 us at one -- the intercept of launch time over the number of channel chunks).  Here the bias quads of the
// wave's couts are loaded ONCE before the first store, the residual quads of a group of four are loaded together, and the
// stores go out back to back.  LeakyReLU / ReLU / none are one select-and-multiply with a per-layer slope; ELU is the
// second instantiation.  Quads that are not fully inside [0, cout) or not 16-byte aligned, and sigmoid layers, fall back
// to conv_epilogue_quad (wave-uniform decision).
template <int NQ>
struct ConvEpi {
    f32x4 b[NQ];
    int col0[NQ];
    float slope;  // LeakyReLU a / none 1 as one formula: v > 0 ? v : v * slope; ReLU selects +0 (as apply_act does)
    bool relu;
    bool fast;    // every quad of every lane of the wave takes the vector path
};
// col0_of(q): first cout of quad q for this lane
template <int NQ, class ColOf>
__device__ __forceinline__ void conv_epi_init(const ConvParams& p, ConvEpi<NQ>& c, ColOf col0_of) {
    bool ok = conv_vec_ok(p) && p.act != ACT_SIGMOID;  // (sigmoid: the one- / two-channel heads, never vector quads)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        c.col0[q] = col0_of(q);
        ok = ok && (c.col0[q] + 3 < p.cout);
    }
    c.fast = __all(ok ? 1 : 0) != 0;
    c.slope = p.act == ACT_LEAKY ? p.act_param : (p.act == ACT_RELU ? 0.f : 1.f);
    c.relu = p.act == ACT_RELU;
    if (c.fast) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) c.b[q] = *reinterpret_cast<const f32x4*>(p.bias + c.col0[q]);
    }
}
template <bool ELU, int NQ, class Get>
__device__ __forceinline__ void conv_epi_row_act(const ConvParams& p, const ConvEpi<NQ>& c, size_t m, Get get) {
    float* d = p.dst + m * p.dst_cs + p.dst_co;
    constexpr int GQ = NQ < 4 ? NQ : 4;  // quads per group: a group's residual loads together, its stores back to back
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += GQ) {
        f32x4 r[GQ];
        if (p.res) {
            const float* rp = p.res + m * p.res_cs + p.res_co;
#pragma unroll
            for (int q = 0; q < GQ; ++q) r[q] = *reinterpret_cast<const f32x4*>(rp + c.col0[q0 + q]);
        }
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            f32x4 x = get(q0 + q) + c.b[q0 + q];
            if (p.res) x += r[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = ELU ? (x[e] > 0.f ? x[e] : p.act_param * expm1f(x[e])) : (x[e] > 0.f ? x[e] : (c.relu ? 0.f : x[e] * c.slope));
            *reinterpret_cast<f32x4*>(d + c.col0[q0 + q]) = x;
        }
    }
}
// one output pixel (linear index m) x NQ quads; get(q) = the accumulated quad; `valid` false = the lane has no pixel here
template <int NQ, class Get>
__device__ __forceinline__ void conv_epi_row(const ConvParams& p, const ConvEpi<NQ>& c, size_t m, bool valid, Get get) {
    if (!c.fast) {
        if (valid) {
            const bool vec_ok = conv_vec_ok(p);
#pragma unroll
            for (int q = 0; q < NQ; ++q) conv_epilogue_quad(p, m, c.col0[q], get(q), vec_ok);
        }
        return;
    }
    if (!valid) return;
    if (p.act == ACT_ELU)  // wave-uniform
        conv_epi_row_act<true, NQ>(p, c, m, get);
    else
        conv_epi_row_act<false, NQ>(p, c, m, get);
}


