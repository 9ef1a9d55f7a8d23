// Host-side executors of the two CNNs: own every device buffer, pack weights once, replay a fixed
// launch sequence on one HIP stream (captured into a hipGraph after the first run).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "dfvo_common.h"

namespace dfvo {

struct HostTensor {
    std::vector<float> data;
    std::vector<int> shape;
};

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
    int alloc(size_t floats);  // zero-filled
    void release();
};

struct ConvLayer {
    float* wp = nullptr;
    float* bias = nullptr;
    unsigned short* wf = nullptr;  // f16 hi/lo planes for conv_win_f16s_kernel (DFVO_CONV_PRECISION=f16x3, 3x3 layers)
    int wf_cout_pad = 0;
    unsigned short* wg = nullptr;  // f16 hi/lo planes in k-group order for conv_gemm_f16s_kernel (f16x3 mode, every layer)
    uint32_t* gtab = nullptr;      // its k-group table
    float* wg32 = nullptr;         // fp32 weights in the same k-group order for conv_gemm_f32g_kernel (fp32 mode, every layer)
    int wg_cout_pad = 0, g_steps = 0;
    long long m_hint = 0;          // output pixels the layer was built for (0: unknown, dfvo_conv2d) -- gates optional packings
    int f16_terms = 3;             // 1: packed in "f16" mode (one product per term; the kernels never read the lo planes)
    float* wh = nullptr;  // head layout (cout <= 2, square 3/5/7 kernels), see conv_pack_head_weights
    int cout = 0, cout_pad = 0, c0 = 0, c1 = 0, kh = 0, kw = 0, ksteps = 0;
    int stride = 1, pad_h = 0, pad_w = 0, pad_mode = PAD_ZERO, act = ACT_NONE;
    float act_param = 0.f;
    double macs_per_pixel() const { return (double)cout * (c0 + c1) * kh * kw; }
};

struct ParamStore {
    std::map<std::string, HostTensor> t;
    const HostTensor* get(const std::string& name) const;
};

// builds + uploads a conv layer from OIHW weights; scale/shift fold an eval BatchNorm
int make_conv(const ParamStore& ps, const std::string& wname, const std::string& bname, int c0, int c1,
              long long M_hint, const float* scale, const float* shift, ConvLayer* out);
void free_conv(ConvLayer* l);
// uploads the head-layout copy of the weights when the layer qualifies for the direct head kernel (else leaves wh null)
int make_f16s_weights(const float* w_oihw, int cout, int c0, int c1, int kh, int kw, const float* scale, ConvLayer* L);
int make_f16g_weights(const float* w_oihw, int cout, int c0, int c1, int kh, int kw, const float* scale, ConvLayer* L);
int conv_set_precision(const char* name);  // fp32 | f16x3 | f16: applies to layers packed afterwards
const char* conv_get_precision();
int make_head_weights(const float* w_oihw, int cout, int c0, int c1, int kh, int kw, const float* scale, float** wh);


struct View {
    const float* p;
    int cs, co;
};

int run_conv(const ConvLayer& L, int N, int H, int W, View s0, int up0, View s1, const float* res, int res_cs,
             int res_co, float* dst, int dst_cs, int dst_co, int dst_zero_to, hipStream_t s, double* flops,
             const DevBuf* splitk_ws = nullptr);

// ------------------------------------------------------------------------------------------------
// flow-net input size for an image size, exactly as the reference's DeepFlow.get_target_size evaluates (nets.hip)
void flow_target_size(int h, int w, int* th, int* tw);

struct FlowNet {
    int imgH = 0, imgW = 0;  // cfg image size
    int H = 0, W = 0;        // net size (multiple of 32), deep_flow.py:89-105
    hipStream_t stream = nullptr;
    bool own_stream = false;
    ParamStore params;
    bool finalized = false;

    // per level l = 1..6 (index l)
    int lh[7], lw[7], lc[7];
    DevBuf img[7], feat[7];
    DevBuf lin_x[7], lin_y[7];
    std::vector<ConvLayer> feat_convs;  // 12 convs of Features
    struct Level {
        ConvLayer m_feat, m_main[4], s_feat, s_main[4], r_feat, r_main[6], r_dist[2];
        bool has_mfeat = false, has_upflow = false, has_upcorr = false, has_rfeat = false, dist_sep = false;
        DevBuf upflow_w, upcorr_w, scale_wx, scale_wy;
        float scale_bx = 0.f, scale_by = 0.f;
        // activations
        DevBuf mfeat, sfeat, rfeat, flow_up, warped, corr, corr_up, x128, x64, x32, x128b, x64b, x32b, flowM, b1,
            flowS, r0, dist_a, dist_b, flow, mean;
    } lv[7];
    DevBuf u8_ref, u8_cur;  // staging for the host-pointer entry point
    DevBuf splitk;          // split-K partial sums of the small-grid convs (own buffer: the nets run on separate streams)
    DevBuf out_fwd, out_bwd, out_diff;
    double flops_last = 0.0;  // useful conv+corr FLOPs of the last forward (2*MAC)
    // ... of its parts: Features on both frames / on the current frame alone (carried mode) / everything behind them
    double flops_feat2 = 0.0, flops_feat1 = 0.0, flops_levels = 0.0;
    struct LevelsGraph {  // the levels, captured per output buffer set
        hipGraph_t g = nullptr;
        hipGraphExec_t exec = nullptr;
        float *fwd = nullptr, *bwd = nullptr, *diff = nullptr;
    } lv_graphs[4];
    int lv_graph_next = 0;
    hipGraph_t graph_feat[2] = {nullptr, nullptr};  // [0] Features of both frames, [1] carry + Features of the current frame
    hipGraphExec_t graph_feat_exec[2] = {nullptr, nullptr};
    const FlowNet* graph_carry_src = nullptr;
    hipEvent_t e_feat = nullptr;  // recorded behind the feature stage of every pass (the carry source of the next pair)
    bool use_graph = true;
    bool tuned_once = false;

    int init(int imgH, int imgW, hipStream_t s);
    int finalize();
    int forward(const uint8_t* d_ref, const uint8_t* d_cur, float* d_fwd, float* d_bwd, float* d_diff,
                const FlowNet* carry_from = nullptr);
    int enqueue_input(const uint8_t* d_ref, const uint8_t* d_cur);  // uint8 frames -> level-1 net input (not captured)
    int enqueue_features(int n0, int N, double* fl);
    int enqueue_features_both();
    int enqueue_carry(const FlowNet& src);
    int enqueue_levels(float* d_fwd, float* d_bwd, float* d_diff);
    int enqueue(float* d_fwd, float* d_bwd, float* d_diff);  // Features of both frames + levels
    void destroy();
};

struct DepthNet {
    int H = 0, W = 0;  // feed size
    hipStream_t stream = nullptr;
    bool own_stream = false;
    ParamStore params;
    bool finalized = false;
    float min_depth = 0.1f, max_depth = 100.f, baseline_mult = 5.4f;

    ConvLayer conv1;
    struct Block {
        ConvLayer c1, c2, ds;
        bool has_ds = false;
    } blocks[8];
    ConvLayer up[5][2], dispconv;
    DevBuf x0, f0, pool, tmp[4], feat[5], blk_t, blk_ds, blk_o[2], du[5], dx[5], disp, depth;
    DevBuf u8_in;
    DevBuf splitk;
    double flops_last = 0.0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    const uint8_t* graph_in = nullptr;
    float* graph_out = nullptr;
    bool use_graph = true;
    bool tuned_once = false;

    int init(int feedH, int feedW, hipStream_t s);
    int finalize();
    int forward(const uint8_t* d_img, float* d_depth);
    int enqueue(const uint8_t* d_img, float* d_depth);
    void destroy();
};

}  // namespace dfvo
