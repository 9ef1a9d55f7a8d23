// Shared internal declarations for libdfvo_hip.so (gfx950 only).
// Activations are NHWC float32; "cs" = floats per pixel of the underlying
// buffer, "co" = channel offset of the view inside the pixel.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/dfvo_hip.h"

namespace dfvo {

void set_last_error(const std::string& s);
const char* last_error();

#define DFVO_HIP_CHECK(expr)                                                        \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            dfvo::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
            return DFVO_ERR_HIP;                                                    \
        }                                                                           \
    } while (0)

#define DFVO_ARG_CHECK(cond, msg)                   \
    do {                                            \
        if (!(cond)) {                              \
            dfvo::set_last_error(std::string(msg)); \
            return DFVO_ERR_ARG;                    \
        }                                           \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device function attribute: remembered per (device, kernel) so that a
// process that drives several GPUs (dfvo_set_device) or several pipelines from different threads configures each one
int ensure_dyn_lds(const void* kernel, size_t bytes);

enum Act { ACT_NONE = 0, ACT_LEAKY = 1, ACT_RELU = 2, ACT_ELU = 3, ACT_SIGMOID = 4 };
enum PadMode { PAD_ZERO = 0, PAD_REFLECT = 1 };

// Implicit-GEMM convolution parameters (device-visible, passed by value).
struct ConvParams {
    // logical input spatial dims the taps index into (after optional x2 nearest upsample of src0)
    int N, H, W;
    int Ho, Wo;
    int kh, kw, stride, pad_h, pad_w;
    int pad_mode;
    // source 0: c0 logical channels, G0 = ceil(c0/4) k-groups per tap
    const float* src0;
    int G0, cs0, co0, up0;
    // source 1 (optional; G1 == 0 when absent)
    const float* src1;
    int G1, cs1, co1;
    // packed weights [ksteps*4][cout_pad][4], bias[cout_pad]
    const float* wp;
    const float* bias;
    // head layout of the same weights (one- / two-channel square layers only, else null): see conv_pack_head_weights
    const float* wh;
    // f16 hi/lo planes of the weights in the split window kernel's layout (DFVO_CONV_PRECISION=f16x3, 3x3 layers only):
    // [tap][16-channel chunk][wf16_cout_pad / 32][plane hi, lo][k / 8][32 couts][8] halves, see conv_pack_weights_f16s
    const unsigned short* wf16;
    int wf16_cout_pad;
    // f16 hi/lo planes in k-group order for the generic split kernel (conv_gemm_f16s.h; every layer in f16x3 mode):
    // [16-k step][wf16g_cout_pad / 32][plane][k / 8][32 couts][8] halves + the layer's k-group table (4 words per step)
    const unsigned short* wf16g;
    int wf16g_cout_pad;
    const uint32_t* f16g_tab;
    int f16g_steps;
    // products per term of the f16 kernels: 3 (or 0) = f16x3 (hi x hi + the two cross products, fp32-class), 1 = "f16" mode
    // (hi x hi only: plain f16 operands, fp32 accumulate -- BASELINE config 5's "fp16 flow"; the lo planes are never read)
    int f16_terms;
    // exact-fp32 twin of the above for conv_gemm_f32g_kernel (fp32 mode): [16-k step][cout_pad / 32][k / 8][32 couts][8]
    // floats, same k-group table and cout padding (wf16g_cout_pad, f16g_tab, f16g_steps are set in both modes)
    const float* wf32g;
    int cout, cout_pad, ksteps;
    // optional residual (added before activation)
    const float* res;
    int res_cs, res_co;
    int act;
    float act_param;
    float* dst;
    int dst_cs, dst_co;
    // when != 0 the epilogue also zero-fills channels [cout, dst_zero_to) of dst
    int dst_zero_to;
    // split-K workspace (optional): partial accumulators [splits][M][cout_pad]
    float* ws;
    size_t ws_floats;
    // split-K tickets (optional, zero between launches): one counter per output tile.  With them the workgroup that writes
    // a tile's last partial also reduces and finishes the tile -- no second launch (see splitk_last_arriver)
    unsigned* tile_flags;
    int tile_flags_n;
    // 2 * MACs of the unpadded convolution (bookkeeping for the bench's roofline leg; not read on device)
    double useful_flops;
    // launch overrides chosen by the per-layer autotuner (0 = heuristic): M tile rows, split-K factor
    int force_bm, force_splits;
    // the device counter of activations beyond f16's range (g_f16s_clamped of conv_win_f16s.h) for the f16 kernels that
    // live in their own translation units (conv_taps_f16s.hip: device symbols do not cross TUs)
    unsigned* f16s_clamp_ctr;
};

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Streams of the pipeline are plain non-blocking streams.  Measured dead ends, kept as records only:
// stream priorities for the solver chain (neutral for the chain, -3 .. -9 % for the others: round 2), CU masks that keep a
// few CUs per XCD free of net kernels or confine the solver to them (-10 %: round 2).
static inline hipError_t create_net_stream(hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
static inline hipError_t create_solver_stream(hipStream_t* s, int = 1) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }

// candidate streams classified by the dispatch pipe that serves them (stream_pool.hip)
struct StreamPool {
    std::vector<hipStream_t> s;
    std::vector<int> group;  // pipe group of s[i] (streams of one group slow each other's dispatch), -1 unknown
    int ngroups = 0;         // 0: the probe failed, use creation order
    int create(int n);
    hipStream_t take(int group);  // removes a stream of that group from the pool (nullptr when none is left)
    int count(int group) const;
    void release();               // destroys what was not taken
};

// number of K-steps (16 k-values = 4 groups of 4 channels) for a conv
static inline int conv_ksteps(int kh, int kw, int c0, int c1) {
    int G = cdiv(c0, 4) + cdiv(c1, 4);
    return cdiv(kh * kw * G, 4);
}
// N-tile (BN) the launcher will use for a given cout; cout_pad is a multiple of it
int conv_pick_bn(int cout, long long M);
static inline int conv_cout_pad(int cout, long long M) {
    int bn = conv_pick_bn(cout, M);
    return round_up(cout, bn);
}
// Pack OIHW weights (cin = c0 + c1) into the kernel layout. `out` must hold
// conv_ksteps*4*cout_pad*4 floats. fold_scale/fold_shift (per-cout, may be null)
// implement eval-mode BatchNorm folding: w' = w*scale, b' = b*scale + shift.
// weights of a one- / two-channel k x k layer in the order the direct head kernel consumes them:
// [8-channel chunk (source 0 first, each source rounded up)][kx][4-channel group of the chunk (2)][ky][cout][4]
int conv_split_mode();  // 0 exact fp32 (default), 4 = f16x3 (f16 hi/lo planes, fp32-class), 5 = f16 (hi plane only: one product per term)
// weights of a 3x3 layer as f16 hi / lo planes for conv_win_f16s_kernel; returns the number of halves written
// (out may be null to query the size): [tap][chunk][cout_pad32][2][16], chunks = ceil(c0/16) + ceil(c1/16)
size_t conv_pack_weights_f16s(const float* w_oihw, int cout, int c0, int c1, const float* fold_scale, unsigned short* out);
unsigned* conv_f16s_overflow_counter();  // device address of the counter behind it (session.hip reads it behind each net)
int conv_f16s_overflow_count(unsigned long long* n, int reset);  // saturation report of the f16x3 split (conv_win_f16s.h)
void conv_build_f16g_table(int c0, int c1, int kh, int kw, std::vector<uint32_t>* tab);
size_t conv_pack_weights_f16g(const float* w_oihw, int cout, int c0, int c1, int kh, int kw, const float* fold_scale,
                              unsigned short* out);
size_t conv_pack_weights_f32g(const float* w_oihw, int cout, int c0, int c1, int kh, int kw, const float* fold_scale, float* out);
bool conv_f32g_ok(const ConvParams& p);
int launch_f32g(const ConvParams& p, hipStream_t stream, int* ksp_out, int* grid_x);
size_t conv_head_weight_floats(int cout, int c0, int c1, int k);
void conv_pack_head_weights(const float* w_oihw, int cout, int c0, int c1, int k, const float* fold_scale, float* out);
void conv_pack_weights(const float* w_oihw, const float* bias, int cout, int c0, int c1, int kh, int kw,
                       int cout_pad, const float* fold_scale, const float* fold_shift, float* out_w,
                       float* out_b);

constexpr int CONV_NUM_CFGS = 24;  // tile configurations of the implicit-GEMM kernel (profile arrays have this size)
int launch_conv(const ConvParams& p, hipStream_t stream);
void conv_profile_begin();
int conv_profile_end(double* ms, double* flops, int* launches, double* bytes = nullptr);

}  // namespace dfvo
