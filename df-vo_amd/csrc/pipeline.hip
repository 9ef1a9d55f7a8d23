// Fused per-pair tracking pipeline: both CNNs, keypoint selection, E-tracker and scale recovery chained
// on the device (three HIP streams), with two host round trips per pair (keypoint count, final pose).
// Orchestration follows /root/reference/libs/dfvo.py:299-345 (deep_model_inference) and :121-262
// (tracking): the host-side glue of the reference (cv2.resize nearest, preprocess_depth, dict passing)
// becomes device kernels; the pose composition stays on the host (dfvo.py:109-119).
#include "../../include/dfvo_hip.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nets.h"
#include "ops.h"
#include "resize_lanczos.h"
#include "tracker.h"

using namespace dfvo;

struct dfvo_pipeline {
    int H = 0, W = 0, feedH = 0, feedW = 0;
    FlowNet flow;
    // further flow-net instances, each on its own stream (DFVO_FLOW_INSTANCES, default 2): consecutive pairs rotate
    // over the instances, so the latency-bound coarse pyramid levels of one pass overlap the throughput-bound fine
    // levels of another (8.7 -> 7.1 ms per pair with two); costs one set of activations (~1.3 GB) per instance
    FlowNet flow_x[DFVO_PIPELINE_SLOTS - 1];
    hipStream_t s_flow_x[DFVO_PIPELINE_SLOTS - 1] = {};
    int flow_instances = 1;
    DepthNet depth;
    TrackerBuffers tbs[DFVO_PIPELINE_SLOTS];  // [0] owns the numpy RandomState and the RNG-side streams, the others share them
    // RNG-independent half of the solver stage (keypoint selection, homography chain) enqueued ahead of time
    // (dfvo_pipeline_prefetch_track): two streams used alternately, per-slot completion event and pinned keypoint info
    hipStream_t s_pre[2] = {nullptr, nullptr};
    hipEvent_t e_pre[DFVO_PIPELINE_SLOTS] = {};
    hipEvent_t e_ref = nullptr;  // reference depth of the first frame written (dfvo_pipeline_set_ref_image / _set_ref_depth)
    int* h_info[DFVO_PIPELINE_SLOTS] = {};
    bool prefetched[DFVO_PIPELINE_SLOTS] = {};
    // dfvo_pipeline_track_begin / _end: results of the RandomState-ordered chain land in pinned host memory behind e_res
    void* h_res[DFVO_PIPELINE_SLOTS] = {};   // PoseState | ScaleResult
    hipEvent_t e_res[DFVO_PIPELINE_SLOTS] = {};
    int begun_n[DFVO_PIPELINE_SLOTS] = {};   // -1: no chain pending, -2: pending pair had no good keypoints, else keypoint count
    const double* begun_depth_override[DFVO_PIPELINE_SLOTS] = {};
    int pending_slot = -1;  // the one pair begun and not yet collected (the chains consume ONE RandomState, in pair order)
    hipStream_t s_flow = nullptr, s_depth = nullptr, s_trk = nullptr;
    hipEvent_t e_flow[DFVO_PIPELINE_SLOTS] = {}, e_depth[DFVO_PIPELINE_SLOTS] = {};
    // the roll-over copy ref_depth <- proc_depth[slot] (s_trk) has read its source / written its target: the depth stream's
    // next writers of either wait for it on the device (recorded by roll_ref_depth; null until the first roll-over)
    hipEvent_t e_roll = nullptr;
    bool roll_pending = false;
    // per-slot outputs of the nets
    float *fwd[DFVO_PIPELINE_SLOTS] = {}, *bwd[DFVO_PIPELINE_SLOTS] = {}, *diff[DFVO_PIPELINE_SLOTS] = {};
    float* raw_depth[DFVO_PIPELINE_SLOTS] = {};
    double* proc_depth[DFVO_PIPELINE_SLOTS] = {};
    float* depth_small = nullptr;
    LanczosResizer feed_resize;  // current frame -> depth-net feed size (when the caller passes no resized frame)
    uint8_t* feed_buf = nullptr;
    double* d_T21 = nullptr;
    // PnP fallback: processed depth of the reference frame (= the previous pair's current frame)
    PnpBuffers pnp;
    double* ref_depth = nullptr;
    float* ref_raw = nullptr;
    bool has_ref_depth = false;
    dfvo_pipeline_cfg cfg;
    bool nets_ready = false;
    const FlowNet* last_flow = nullptr;  // the instance that ran the previous pair (carry source of a d_ref == NULL call)
};

#define P_TRY(expr)                     \
    do {                                \
        int _rc = (expr);               \
        if (_rc != DFVO_OK) return _rc; \
    } while (0)

extern "C" {

int dfvo_pipeline_create(const dfvo_pipeline_cfg* cfg, dfvo_pipeline** out) {
    DFVO_ARG_CHECK(cfg && out, "dfvo_pipeline_create: null argument");
    dfvo_pipeline* p = new dfvo_pipeline();
    p->cfg = *cfg;
    p->H = cfg->img_h;
    p->W = cfg->img_w;
    p->feedH = cfg->feed_h;
    p->feedW = cfg->feed_w;
    auto fail = [&](int rc) {
        delete p;
        return rc;
    };
    // Streams by dispatch pipe (stream_pool.hip): twelve candidates, classified by measurement; pipe A / B: one flow-net
    // instance each, pipe C: the depth net + the two run-ahead homography chains, pipe D: the RandomState-ordered chain and
    // its two side streams, alone.  Falls back to creation order when the probe does not find four groups of three.
    hipStream_t pool_rep[2] = {nullptr, nullptr}, pool_pre[2] = {nullptr, nullptr}, pool_fx = nullptr;
    {
        StreamPool pool;
        if (pool.create(12) == DFVO_OK && pool.ngroups >= 4) {
            int g[4] = {-1, -1, -1, -1}, ng = 0;  // the four largest groups, largest first
            std::vector<int> order;
            for (int i = 0; i < pool.ngroups; ++i) order.push_back(i);
            std::sort(order.begin(), order.end(), [&](int a, int b) { return pool.count(a) > pool.count(b); });
            for (int i = 0; i < 4; ++i) g[ng++] = order[i];
            // role -> pipe: trk rep0 rep1 | depth pre0 pre1 | flow | flow_x.  Other placements were measured
            // (profiles/r3k_layouts.txt): a run-ahead homography chain on a flow net's pipe 178-194 pairs/s against 266 -- its
            // long single-workgroup kernels hold up the dispatch of the flow net's hundred short launches per pass.
            static const int R[8] = {0, 0, 0, 1, 1, 1, 2, 3};
            if (pool.count(g[0]) >= 3 && pool.count(g[1]) >= 3 && pool.count(g[2]) >= 3 && pool.count(g[3]) >= 3) {
                p->s_trk = pool.take(g[R[0]]);
                pool_rep[0] = pool.take(g[R[1]]);
                pool_rep[1] = pool.take(g[R[2]]);
                p->s_depth = pool.take(g[R[3]]);
                pool_pre[0] = pool.take(g[R[4]]);
                pool_pre[1] = pool.take(g[R[5]]);
                p->s_flow = pool.take(g[R[6]]);
                pool_fx = pool.take(g[R[7]]);
            } else if (pool.count(g[0]) >= 3 && pool.count(g[1]) >= 3 && pool.count(g[2]) >= 1 && pool.count(g[3]) >= 1) {
                p->s_trk = pool.take(g[0]);
                pool_rep[0] = pool.take(g[0]);
                pool_rep[1] = pool.take(g[0]);
                p->s_depth = pool.take(g[1]);
                pool_pre[0] = pool.take(g[1]);
                pool_pre[1] = pool.take(g[1]);
                p->s_flow = pool.take(g[2]);
                pool_fx = pool.take(g[3]);
            }
        }
        pool.release();
    }
    if (!p->s_trk && (create_net_stream(&p->s_flow) != hipSuccess || create_net_stream(&p->s_depth) != hipSuccess ||
                      create_solver_stream(&p->s_trk) != hipSuccess)) {
        dfvo::set_last_error("dfvo_pipeline_create: hipStreamCreate failed (no GPU?)");
        return fail(DFVO_ERR_HIP);
    }
    for (int i = 0; i < DFVO_PIPELINE_SLOTS; i++) {
        if (hipEventCreateWithFlags(&p->e_flow[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p->e_depth[i], hipEventDisableTiming) != hipSuccess) {
            dfvo::set_last_error("dfvo_pipeline_create: hipEventCreate failed");
            return fail(DFVO_ERR_HIP);
        }
    }
    int rc = p->flow.init(p->H, p->W, p->s_flow);
    if (rc != DFVO_OK) return fail(rc);
    // two LiteFlowNet instances on two pipes, alternating pairs (measured 1 / 2 / 3: 217 / 287 / 272 pairs/s, profiles/
    // r3ag_flow_instances_ab.txt).  DFVO_FLOW_INSTANCES=1 is a test hook: the carry-over from a pass of the SAME instance
    p->flow_instances = getenv("DFVO_FLOW_INSTANCES") && atoi(getenv("DFVO_FLOW_INSTANCES")) == 1 ? 1 : 2;
    for (int i = 0; i + 1 < p->flow_instances; ++i) {
        if (i == 0 && pool_fx) {
            p->s_flow_x[0] = pool_fx;
            pool_fx = nullptr;
        } else if (create_net_stream(&p->s_flow_x[i]) != hipSuccess) {
            return fail(DFVO_ERR_HIP);
        }
        rc = p->flow_x[i].init(p->H, p->W, p->s_flow_x[i]);
        if (rc != DFVO_OK) return fail(rc);
    }
    rc = p->depth.init(p->feedH, p->feedW, p->s_depth);
    if (rc != DFVO_OK) return fail(rc);
    p->depth.min_depth = cfg->net_min_depth;
    p->depth.max_depth = cfg->net_max_depth;
    p->depth.baseline_mult = cfg->baseline_mult;
    if (pool_fx) (void)hipStreamDestroy(pool_fx);
    rc = p->tbs[0].init(pool_rep[0], pool_rep[1]);
    if (rc != DFVO_OK) return fail(rc);
    for (int i = 1; i < DFVO_PIPELINE_SLOTS; i++) {
        rc = p->tbs[i].init_shared(p->tbs[0]);
        if (rc != DFVO_OK) return fail(rc);
    }
    // the PnP fallback's buffers at their final size: grown lazily they would be freed and re-allocated (hipFree waits for
    // the whole device) whenever a pair with more keypoints than any before takes the fallback
    if (cfg->kp_num_bestN > 0 && cfg->pnp_iters > 0) {
        rc = p->pnp.ensure(cfg->kp_num_bestN + 8, cfg->pnp_iters);
        if (rc != DFVO_OK) return fail(rc);
    }
    for (int i = 0; i < 2; i++) {
        if (pool_pre[i])
            p->s_pre[i] = pool_pre[i];
        else if (create_solver_stream(&p->s_pre[i], 4) != hipSuccess)
            return fail(DFVO_ERR_HIP);
    }
    for (int i = 0; i < DFVO_PIPELINE_SLOTS; i++) {
        if (hipEventCreateWithFlags(&p->e_pre[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p->e_res[i], hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc((void**)&p->h_info[i], 4 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void**)&p->h_res[i], sizeof(PoseState) + sizeof(ScaleResult), hipHostMallocDefault) != hipSuccess)
            return fail(DFVO_ERR_HIP);
        p->begun_n[i] = -1;
    }
    const size_t px = (size_t)p->H * p->W;
    for (int i = 0; i < DFVO_PIPELINE_SLOTS; i++) {
        if (hipMalloc((void**)&p->fwd[i], 2 * px * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&p->bwd[i], 2 * px * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&p->diff[i], px * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&p->raw_depth[i], px * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&p->proc_depth[i], px * sizeof(double)) != hipSuccess) {
            dfvo::set_last_error("dfvo_pipeline_create: hipMalloc failed");
            return fail(DFVO_ERR_HIP);
        }
    }
    if (hipMalloc((void**)&p->depth_small, (size_t)p->feedH * p->feedW * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&p->ref_depth, px * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&p->ref_raw, px * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&p->d_T21, 16 * sizeof(double)) != hipSuccess) {
        dfvo::set_last_error("dfvo_pipeline_create: hipMalloc failed");
        return fail(DFVO_ERR_HIP);
    }
    if (hipMalloc((void**)&p->feed_buf, (size_t)p->feedH * p->feedW * 3) != hipSuccess) {
        dfvo::set_last_error("dfvo_pipeline_create: hipMalloc failed");
        return fail(DFVO_ERR_HIP);
    }
    if (p->feed_resize.init(p->H, p->W, p->feedH, p->feedW) != DFVO_OK) return fail(DFVO_ERR_HIP);
    enqueue_mt_seed(p->tbs[0], cfg->seed, p->s_trk);
    (void)hipStreamSynchronize(p->s_trk);
    *out = p;
    return DFVO_OK;
}

void dfvo_pipeline_destroy(dfvo_pipeline* p) {
    if (!p) return;
    (void)hipDeviceSynchronize();
    p->flow.destroy();
    for (int i = 0; i + 1 < p->flow_instances; ++i) {
        p->flow_x[i].destroy();
        if (p->s_flow_x[i]) (void)hipStreamDestroy(p->s_flow_x[i]);
    }
    p->depth.destroy();
    for (int i = DFVO_PIPELINE_SLOTS - 1; i >= 0; i--) p->tbs[i].release();
    for (int i = 0; i < 2; i++)
        if (p->s_pre[i]) (void)hipStreamDestroy(p->s_pre[i]);
    for (int i = 0; i < DFVO_PIPELINE_SLOTS; i++) {
        if (p->e_pre[i]) (void)hipEventDestroy(p->e_pre[i]);
        if (p->e_res[i]) (void)hipEventDestroy(p->e_res[i]);
        if (p->h_res[i]) (void)hipHostFree(p->h_res[i]);
        if (i == 0 && p->e_ref) (void)hipEventDestroy(p->e_ref);
        if (p->h_info[i]) (void)hipHostFree(p->h_info[i]);
    }
    for (int i = 0; i < DFVO_PIPELINE_SLOTS; i++) {
        void* ptrs[] = {p->fwd[i], p->bwd[i], p->diff[i], p->raw_depth[i], p->proc_depth[i]};
        for (void* q : ptrs)
            if (q) (void)hipFree(q);
        if (p->e_flow[i]) (void)hipEventDestroy(p->e_flow[i]);
        if (p->e_depth[i]) (void)hipEventDestroy(p->e_depth[i]);
        if (i == 0 && p->e_roll) (void)hipEventDestroy(p->e_roll);
    }
    if (p->depth_small) (void)hipFree(p->depth_small);
    p->feed_resize.release();
    if (p->feed_buf) (void)hipFree(p->feed_buf);
    if (p->ref_depth) (void)hipFree(p->ref_depth);
    if (p->ref_raw) (void)hipFree(p->ref_raw);
    p->pnp.release();
    if (p->d_T21) (void)hipFree(p->d_T21);
    if (p->s_flow) (void)hipStreamDestroy(p->s_flow);
    if (p->s_depth) (void)hipStreamDestroy(p->s_depth);
    if (p->s_trk) (void)hipStreamDestroy(p->s_trk);
    delete p;
}

static int pipe_store(ParamStore* ps, const char* name, const float* h, int ndim, const int* shape) {
    DFVO_ARG_CHECK(name && h && ndim >= 0 && ndim <= 8, "set_param: bad argument");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(h, h + n);
    ps->t[name] = std::move(t);
    return DFVO_OK;
}

int dfvo_pipeline_set_flow_param(dfvo_pipeline* p, const char* name, const float* h, int ndim, const int* shape) {
    DFVO_ARG_CHECK(p && !p->nets_ready, "dfvo_pipeline_set_flow_param: bad state");
    for (int i = 0; i + 1 < p->flow_instances; ++i) P_TRY(pipe_store(&p->flow_x[i].params, name, h, ndim, shape));
    return pipe_store(&p->flow.params, name, h, ndim, shape);
}
int dfvo_pipeline_set_depth_param(dfvo_pipeline* p, const char* name, const float* h, int ndim, const int* shape) {
    DFVO_ARG_CHECK(p && !p->nets_ready, "dfvo_pipeline_set_depth_param: bad state");
    return pipe_store(&p->depth.params, name, h, ndim, shape);
}
int dfvo_pipeline_finalize(dfvo_pipeline* p) {
    DFVO_ARG_CHECK(p, "null pipeline");
    P_TRY(p->flow.finalize());
    for (int i = 0; i + 1 < p->flow_instances; ++i) P_TRY(p->flow_x[i].finalize());
    P_TRY(p->depth.finalize());
    p->nets_ready = true;
    return DFVO_OK;
}
int dfvo_pipeline_set_graph(dfvo_pipeline* p, int enable) {
    DFVO_ARG_CHECK(p, "null pipeline");
    p->flow.use_graph = enable != 0;
    for (int i = 0; i + 1 < p->flow_instances; ++i) p->flow_x[i].use_graph = enable != 0;
    p->depth.use_graph = enable != 0;
    return DFVO_OK;
}
int dfvo_pipeline_seed(dfvo_pipeline* p, uint32_t seed) {
    DFVO_ARG_CHECK(p, "null pipeline");
    // synchronous: the first RandomState consumer of the next pair (the keypoint shuffles) runs on tb.s_rep[0], which is
    // ordered after the keypoint stage but not after s_trk -- the new key must be in place before track() is called
    P_TRY(enqueue_mt_seed(p->tbs[0], seed, p->s_trk));
    DFVO_HIP_CHECK(hipStreamSynchronize(p->s_trk));
    return DFVO_OK;
}

int dfvo_pipeline_enqueue_nets(dfvo_pipeline* p, int slot, const uint8_t* d_ref, const uint8_t* d_cur,
                               const uint8_t* d_cur_feed) {
    DFVO_ARG_CHECK(p && p->nets_ready && (slot >= 0 && slot < DFVO_PIPELINE_SLOTS) && d_cur,
                   "dfvo_pipeline_enqueue_nets: bad argument");
    DFVO_ARG_CHECK(d_ref || p->last_flow, "dfvo_pipeline_enqueue_nets: d_ref == NULL (reference frame = the previous call's "
                                          "current frame) needs a previous call");
    const size_t px = (size_t)p->H * p->W;
    // depth of the current frame (dfvo.py:305-319); without a caller-resized frame the LANCZOS resize of
    // deep_models.py:195-199 runs here, ahead of the net on its stream
    if (!d_cur_feed) {
        P_TRY(p->feed_resize.enqueue(d_cur, p->feed_buf, p->s_depth));
        d_cur_feed = p->feed_buf;
    }
    P_TRY(p->depth.forward(d_cur_feed, p->depth_small));
    const dfvo_pipeline_cfg& c = p->cfg;
    const int y0 = (int)(p->H * c.depth_crop[0]), y1 = (int)(p->H * c.depth_crop[1]);
    const int x0 = (int)(p->W * c.depth_crop[2]), x1 = (int)(p->W * c.depth_crop[3]);
    // proc_depth[slot] may still be the source of the previous occupant's roll-over copy on s_trk (track_end does not
    // wait for it on the host): order the overwrite behind it.  Placed after the net, so nothing stalls in practice.
    if (p->roll_pending) DFVO_HIP_CHECK(hipStreamWaitEvent(p->s_depth, p->e_roll, 0));
    P_TRY(launch_depth_post(p->depth_small, p->feedH, p->feedW, p->H, p->W, y0, y1, x0, x1, (float)c.min_depth,
                            (float)c.max_depth, p->raw_depth[slot], p->proc_depth[slot], p->s_depth));
    DFVO_HIP_CHECK(hipEventRecord(p->e_depth[slot], p->s_depth));
    // forward/backward flow (dfvo.py:321-335)
    const int inst = slot % p->flow_instances;
    FlowNet& fn = inst == 0 ? p->flow : p->flow_x[inst - 1];
    hipStream_t sf = fn.stream;
    // frame pointers may change per pair (not captured).  d_ref == NULL: the image / feature pyramids of the reference frame
    // are carried over from the pass that saw it as its current frame (the other flow-net instance, normally)
    // Whatever this pass is (carried or not), it overwrites this instance's pyramids: the previous pass -- normally on the
    // OTHER instance -- may still be copying them (its carry-over reads this instance's current-frame pyramids), so this
    // stream is ordered behind that pass's feature stage.  A carried pass waits for the same event anyway; a full pass
    // enqueued right behind a carried one without a sync in between would otherwise race with that copy.
    if (p->last_flow && p->last_flow != &fn && p->last_flow->e_feat) DFVO_HIP_CHECK(hipStreamWaitEvent(sf, p->last_flow->e_feat, 0));
    // (the net writing the slot's buffers itself, one levels graph per slot, was measured: no gain -- profiles/r3x_copy_ab.txt)
    P_TRY(fn.forward(d_ref, d_cur, fn.out_fwd.p, fn.out_bwd.p, fn.out_diff.p, d_ref ? nullptr : p->last_flow));
    DFVO_HIP_CHECK(hipMemcpyAsync(p->fwd[slot], fn.out_fwd.p, 2 * px * sizeof(float), hipMemcpyDeviceToDevice, sf));
    DFVO_HIP_CHECK(hipMemcpyAsync(p->bwd[slot], fn.out_bwd.p, 2 * px * sizeof(float), hipMemcpyDeviceToDevice, sf));
    DFVO_HIP_CHECK(hipMemcpyAsync(p->diff[slot], fn.out_diff.p, px * sizeof(float), hipMemcpyDeviceToDevice, sf));
    p->last_flow = &fn;
    DFVO_HIP_CHECK(hipEventRecord(p->e_flow[slot], sf));
    return DFVO_OK;
}

int dfvo_pipeline_set_ref_depth(dfvo_pipeline* p, const uint8_t* d_feed, const double* d_depth_override) {
    DFVO_ARG_CHECK(p && p->nets_ready && ((d_feed != nullptr) != (d_depth_override != nullptr)),
                   "dfvo_pipeline_set_ref_depth: exactly one of d_feed / d_depth_override");
    const size_t px = (size_t)p->H * p->W;
    if (d_depth_override) {
        DFVO_HIP_CHECK(hipMemcpyAsync(p->ref_depth, d_depth_override, px * sizeof(double), hipMemcpyDeviceToDevice, p->s_trk));
        DFVO_HIP_CHECK(hipStreamSynchronize(p->s_trk));
    } else {
        P_TRY(p->depth.forward(d_feed, p->depth_small));
        const dfvo_pipeline_cfg& c = p->cfg;
        const int y0 = (int)(p->H * c.depth_crop[0]), y1 = (int)(p->H * c.depth_crop[1]);
        const int x0 = (int)(p->W * c.depth_crop[2]), x1 = (int)(p->W * c.depth_crop[3]);
        if (p->roll_pending) DFVO_HIP_CHECK(hipStreamWaitEvent(p->s_depth, p->e_roll, 0));  // WAW on ref_depth vs the roll-over
        P_TRY(launch_depth_post(p->depth_small, p->feedH, p->feedW, p->H, p->W, y0, y1, x0, x1, (float)c.min_depth,
                                (float)c.max_depth, p->ref_raw, p->ref_depth, p->s_depth));
        // not waited for on the host: the solver stream (PnP fallback reads the reference depth, the roll-over writes it)
        // is ordered behind it on the device, the depth stream runs its later passes in order anyway
        if (!p->e_ref) DFVO_HIP_CHECK(hipEventCreateWithFlags(&p->e_ref, hipEventDisableTiming));
        DFVO_HIP_CHECK(hipEventRecord(p->e_ref, p->s_depth));
        DFVO_HIP_CHECK(hipStreamWaitEvent(p->s_trk, p->e_ref, 0));
    }
    p->has_ref_depth = true;
    return DFVO_OK;
}

int dfvo_pipeline_set_ref_image(dfvo_pipeline* p, const uint8_t* d_img) {
    DFVO_ARG_CHECK(p && p->nets_ready && d_img, "dfvo_pipeline_set_ref_image: bad argument");
    P_TRY(p->feed_resize.enqueue(d_img, p->feed_buf, p->s_depth));
    return dfvo_pipeline_set_ref_depth(p, p->feed_buf, nullptr);
}

// the current frame's depth becomes the reference depth of the next pair (dfvo.py:  ref_data <- cur_data)
static int roll_ref_depth(dfvo_pipeline* p, int slot, const double* d_depth_override) {
    const size_t px = (size_t)p->H * p->W;
    DFVO_HIP_CHECK(hipStreamWaitEvent(p->s_trk, p->e_depth[slot], 0));
    const double* depth = d_depth_override ? d_depth_override : p->proc_depth[slot];
    DFVO_HIP_CHECK(hipMemcpyAsync(p->ref_depth, depth, px * sizeof(double), hipMemcpyDeviceToDevice, p->s_trk));
    if (!p->e_roll) DFVO_HIP_CHECK(hipEventCreateWithFlags(&p->e_roll, hipEventDisableTiming));
    DFVO_HIP_CHECK(hipEventRecord(p->e_roll, p->s_trk));
    p->roll_pending = true;
    p->has_ref_depth = true;
    return DFVO_OK;
}

static void fill_pose_cfg(const dfvo_pipeline_cfg& c, PoseConfig* pc) {
    pc->fx = c.fx;
    pc->cx = c.cx;
    pc->cy = c.cy;
    pc->reproj_thre = c.e_reproj_thre;
    pc->repeat = c.e_repeat;
    pc->max_iters = c.e_max_iters;
    for (int i = 0; i < 9; i++) {
        pc->KinvT[i] = c.KinvT[i];
        pc->Kinv[i] = c.Kinv[i];
    }
}

// RNG-independent half of the solver stage of `slot` (keypoint selection, homography RANSAC + refinement, GRIC-H):
// waits on the device for the slot's flow outputs and runs on a stream of its own, so it executes as soon as those
// nets are done -- typically while dfvo_pipeline_track of the previous pair is still blocking the host.  The numpy
// RandomState is not touched here; all RNG consumers stay in dfvo_pipeline_track, in pair order.
static int enqueue_pre_part(dfvo_pipeline* p, int slot, const float* d_flow_override, const float* d_diff_override,
                            hipStream_t sp) {
    const dfvo_pipeline_cfg& c = p->cfg;
    TrackerBuffers& tb = p->tbs[slot];
    DFVO_HIP_CHECK(hipStreamWaitEvent(sp, p->e_flow[slot], 0));
    const float* flow = d_flow_override ? d_flow_override : p->fwd[slot];
    const float* diff = d_diff_override ? d_diff_override : p->diff[slot];
    P_TRY(enqueue_local_bestn(tb, flow, diff, p->H, p->W, c.kp_num_row, c.kp_num_col, c.kp_num_bestN, (float)c.kp_thre, sp));
    DFVO_HIP_CHECK(hipMemcpyAsync(p->h_info[slot], tb.kp_info, 3 * sizeof(int), hipMemcpyDeviceToHost, sp));
    DFVO_HIP_CHECK(hipEventRecord(p->e_pre[slot], sp));  // the host only needs the keypoint count; tb.ev_h orders the rest
    PoseConfig pc;
    fill_pose_cfg(c, &pc);
    P_TRY(enqueue_pose_h_part(tb, tb.kp_cap, pc, sp));  // keypoint count read on the device; kp_cap bounds the launches
    return DFVO_OK;
}

int dfvo_pipeline_prefetch_track(dfvo_pipeline* p, int slot, const float* d_flow_override, const float* d_diff_override) {
    DFVO_ARG_CHECK(p && (slot >= 0 && slot < DFVO_PIPELINE_SLOTS), "dfvo_pipeline_prefetch_track: bad argument");
    DFVO_ARG_CHECK(!p->prefetched[slot], "dfvo_pipeline_prefetch_track: slot already prefetched and not yet tracked");
    P_TRY(enqueue_pre_part(p, slot, d_flow_override, d_diff_override, p->s_pre[slot & 1]));
    p->prefetched[slot] = true;
    return DFVO_OK;
}

// First half of dfvo_pipeline_track: waits for the slot's keypoint stage, enqueues the RandomState-ordered chain (shuffles,
// 5 x five-point RANSAC, GRIC, recoverPose, scale recovery) and the copy of its results into pinned host memory, returns.
// The host is then free to enqueue the next pairs' nets while the chain runs (dfvo_pipeline_track_end collects).
int dfvo_pipeline_track_begin(dfvo_pipeline* p, int slot, const float* d_flow_override, const float* d_diff_override,
                              const double* d_depth_override) {
    DFVO_ARG_CHECK(p && (slot >= 0 && slot < DFVO_PIPELINE_SLOTS), "dfvo_pipeline_track_begin: bad argument");
    DFVO_ARG_CHECK(p->begun_n[slot] == -1, "dfvo_pipeline_track_begin: the slot's previous pair was not collected (track_end)");
    DFVO_ARG_CHECK(p->pending_slot == -1, "dfvo_pipeline_track_begin: another pair is begun and not yet collected -- the PnP decision "
                                          "of track_end (RandomState draws) comes before the next pair's chain");
    const dfvo_pipeline_cfg& c = p->cfg;
    hipStream_t s = p->s_trk;
    TrackerBuffers& tb = p->tbs[slot];
    static const bool trace = getenv("DFVO_TRACK_TRACE") != nullptr;
    if (trace && !tb.ev_t[0])
        for (int i = 0; i < 4; i++) DFVO_HIP_CHECK(hipEventCreate(&tb.ev_t[i]));
    P_TRY(enqueue_scale_prepare(tb, p->H, p->W));  // side stream: the scale stage's fills leave the dependent chain
    if (!p->prefetched[slot]) P_TRY(enqueue_pre_part(p, slot, d_flow_override, d_diff_override, p->s_pre[slot & 1]));
    p->prefetched[slot] = false;
    DFVO_HIP_CHECK(hipEventSynchronize(p->e_pre[slot]));  // keypoint info is in pinned host memory now
    const int* info = p->h_info[slot];
    p->begun_depth_override[slot] = d_depth_override;
    if (!info[1]) {
        p->begun_n[slot] = -2;
        p->pending_slot = slot;
        return DFVO_OK;
    }
    const int n = info[0];
    PoseConfig pc;
    fill_pose_cfg(c, &pc);
    P_TRY(enqueue_pose_e_part(tb, n, pc, s, p->d_T21));  // waits for tb.ev_h (the prefetched half) on the device
    DFVO_HIP_CHECK(hipStreamWaitEvent(s, p->e_depth[slot], 0));
    ScaleConfig sc;
    sc.cx = c.cx;
    sc.cy = c.cy;
    sc.fx = c.fx;
    sc.fy = c.fy;
    sc.min_samples = c.scale_min_samples;
    sc.max_trials = c.scale_max_trials;
    sc.stop_prob = c.scale_stop_prob;
    sc.thre = c.scale_thre;
    const double* depth = d_depth_override ? d_depth_override : p->proc_depth[slot];
    P_TRY(enqueue_find_scale(tb, n, p->d_T21, depth, p->H, p->W, sc, s, tb.pose, true));
    if (tb.ev_t[3]) DFVO_HIP_CHECK(hipEventRecord(tb.ev_t[3], s));
    char* hr = (char*)p->h_res[slot];
    DFVO_HIP_CHECK(hipMemcpyAsync(hr, tb.pose, sizeof(PoseState), hipMemcpyDeviceToHost, s));
    DFVO_HIP_CHECK(hipMemcpyAsync(hr + sizeof(PoseState), tb.scale_out, sizeof(ScaleResult), hipMemcpyDeviceToHost, s));
    DFVO_HIP_CHECK(hipEventRecord(p->e_res[slot], s));
    p->begun_n[slot] = n;
    p->pending_slot = slot;
    return DFVO_OK;
}

// Second half: waits for the chain's results, runs the PnP fallback where the reference takes it (dfvo.py:225-250; decided
// on the host, so the NEXT pair's chain must not be begun before this returns -- it consumes the same RandomState), rolls
// the reference depth over.
int dfvo_pipeline_track_end(dfvo_pipeline* p, int slot, dfvo_track_out* out) {
    DFVO_ARG_CHECK(p && out && (slot >= 0 && slot < DFVO_PIPELINE_SLOTS), "dfvo_pipeline_track_end: bad argument");
    DFVO_ARG_CHECK(p->begun_n[slot] != -1, "dfvo_pipeline_track_end: no pair pending in this slot (track_begin)");
    const dfvo_pipeline_cfg& c = p->cfg;
    hipStream_t s = p->s_trk;
    TrackerBuffers& tb = p->tbs[slot];
    static const bool trace = getenv("DFVO_TRACK_TRACE") != nullptr;  // host-side phase timing (tuning aid)
    static double tr_acc[2] = {0, 0}, tr_dev[3] = {0, 0, 0};
    static int tr_dev_n = 0, tr_n = 0;
    const auto tr0 = std::chrono::steady_clock::now();
    auto tr_ms = [&](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    const int n = p->begun_n[slot];
    const double* d_depth_override = p->begun_depth_override[slot];
    p->begun_n[slot] = -1;
    p->pending_slot = -1;
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < 3; i++) out->R[i * 4] = 1.0;
    const int* info = p->h_info[slot];
    out->n_kp = info[0];
    out->good_kp_found = info[1];
    if (n == -2) {
        out->status = DFVO_TRACK_CONSTANT_MOTION;
        P_TRY(roll_ref_depth(p, slot, d_depth_override));
        DFVO_HIP_CHECK(hipStreamSynchronize(s));
        return DFVO_OK;
    }
    DFVO_HIP_CHECK(hipEventSynchronize(p->e_res[slot]));
    const double tr_wait = tr_ms(tr0);
    PoseState ps;
    ScaleResult sr;
    memcpy(&ps, p->h_res[slot], sizeof(ps));
    memcpy(&sr, (const char*)p->h_res[slot] + sizeof(PoseState), sizeof(sr));
    for (int i = 0; i < 9; i++) out->R[i] = ps.R[i];
    for (int i = 0; i < 3; i++) out->t[i] = ps.t[i];
    out->best_inlier_cnt = ps.best_cnt;
    out->num_valid = ps.num_valid;
    out->cheirality = ps.cheirality;
    const bool t_zero = ps.t[0] == 0 && ps.t[1] == 0 && ps.t[2] == 0;
    out->scale = t_zero ? 0.0 : sr.scale;  // dfvo.py:198: scale recovery only when ||t|| != 0
    out->scale_n_valid = sr.n_valid;
    out->scale_n_trials = sr.n_trials;
    out->scale_n_inliers = sr.n_inliers;
    if (t_zero || sr.scale == -1.0) {  // dfvo.py:225-250: PnP on (ref keypoints + ref depth) -> cur keypoints
        out->status = DFVO_TRACK_NEEDS_PNP;
        if (p->has_ref_depth) {
            PnpConfig pc3;
            pc3.fx = c.fx;
            pc3.fy = c.fy;
            pc3.cx = c.cx;
            pc3.cy = c.cy;
            for (int i = 0; i < 9; i++) pc3.inv_K[i] = c.Kinv[i];
            pc3.min_depth = c.min_depth;
            pc3.max_depth = c.max_depth;
            pc3.repeat = c.pnp_repeat;
            pc3.iters = c.pnp_iters;
            pc3.reproj_thre = c.pnp_reproj_thre;
            P_TRY(enqueue_compute_pose_3d2d(p->pnp, tb.mt_state, tb.kp_ref, tb.kp_cur, tb.kp_info, n,
                                            p->ref_depth, p->H, p->W, pc3, s));
            PnpResult pr;
            DFVO_HIP_CHECK(hipMemcpyAsync(&pr, p->pnp.result, sizeof(pr), hipMemcpyDeviceToHost, s));
            DFVO_HIP_CHECK(hipStreamSynchronize(s));
            DFVO_ARG_CHECK(pr.status >= 0, "dfvo_pipeline_track: the PnP fallback reported an internal error");
            for (int i = 0; i < 9; i++) out->R[i] = pr.R[i];
            for (int i = 0; i < 3; i++) out->t[i] = pr.tvec[i];
            out->scale = 1.0;
            out->pnp_found = pr.found;
            out->pnp_inliers = pr.best_inliers;
            out->pnp_n_filtered = pr.n_filtered;
            out->status = DFVO_TRACK_PNP;
        }
    } else {
        out->status = DFVO_TRACK_E;
    }
    // the roll-over copy is ordered on s_trk ahead of anything the next pair enqueues there: no host wait needed
    P_TRY(roll_ref_depth(p, slot, d_depth_override));
    if (trace) {
        float d01 = 0, d12 = 0, d23 = 0;  // device time of the chain's three segments (valid when the pair took the E path)
        if (n > 10 && hipEventElapsedTime(&d01, tb.ev_t[0], tb.ev_t[1]) == hipSuccess &&
            hipEventElapsedTime(&d12, tb.ev_t[1], tb.ev_t[2]) == hipSuccess &&
            hipEventElapsedTime(&d23, tb.ev_t[2], tb.ev_t[3]) == hipSuccess) {
            tr_dev[0] += d01;
            tr_dev[1] += d12;
            tr_dev[2] += d23;
            tr_dev_n++;
        }
        if (tr_dev_n == 20) {
            fprintf(stderr, "track device ms: shuffles + five-point batch %.3f | bookkeeping + recoverPose %.3f | scale %.3f\n",
                    tr_dev[0] / 20, tr_dev[1] / 20, tr_dev[2] / 20);
            tr_dev[0] = tr_dev[1] = tr_dev[2] = 0;
            tr_dev_n = 0;
        }
        tr_acc[0] += tr_wait;
        tr_acc[1] += tr_ms(tr0);
        if (++tr_n % 20 == 0) {
            fprintf(stderr, "track_end host ms: waited for the chain %.3f | total %.3f\n", tr_acc[0] / 20, tr_acc[1] / 20);
            tr_acc[0] = tr_acc[1] = 0;
        }
    }
    return DFVO_OK;
}

int dfvo_pipeline_track(dfvo_pipeline* p, int slot, const float* d_flow_override, const float* d_diff_override,
                        const double* d_depth_override, dfvo_track_out* out) {
    DFVO_ARG_CHECK(p && out && (slot >= 0 && slot < DFVO_PIPELINE_SLOTS), "dfvo_pipeline_track: bad argument");
    P_TRY(dfvo_pipeline_track_begin(p, slot, d_flow_override, d_diff_override, d_depth_override));
    P_TRY(dfvo_pipeline_track_end(p, slot, out));
    DFVO_HIP_CHECK(hipStreamSynchronize(p->s_trk));  // (synchronous, as documented: the roll-over copy included)
    return DFVO_OK;
}

int dfvo_pipeline_get_flow(dfvo_pipeline* p, int slot, float* h_fwd, float* h_bwd, float* h_diff, float* h_raw_depth,
                           double* h_depth) {
    DFVO_ARG_CHECK(p && (slot >= 0 && slot < DFVO_PIPELINE_SLOTS), "dfvo_pipeline_get_flow: bad argument");
    const size_t px = (size_t)p->H * p->W;
    DFVO_HIP_CHECK(hipDeviceSynchronize());
    if (h_fwd) DFVO_HIP_CHECK(hipMemcpy(h_fwd, p->fwd[slot], 2 * px * sizeof(float), hipMemcpyDeviceToHost));
    if (h_bwd) DFVO_HIP_CHECK(hipMemcpy(h_bwd, p->bwd[slot], 2 * px * sizeof(float), hipMemcpyDeviceToHost));
    if (h_diff) DFVO_HIP_CHECK(hipMemcpy(h_diff, p->diff[slot], px * sizeof(float), hipMemcpyDeviceToHost));
    if (h_raw_depth) DFVO_HIP_CHECK(hipMemcpy(h_raw_depth, p->raw_depth[slot], px * sizeof(float), hipMemcpyDeviceToHost));
    if (h_depth) DFVO_HIP_CHECK(hipMemcpy(h_depth, p->proc_depth[slot], px * sizeof(double), hipMemcpyDeviceToHost));
    return DFVO_OK;
}

int dfvo_pipeline_get_keypoints(dfvo_pipeline* p, int slot, int cap, double* h_kp_ref, double* h_kp_cur,
                                uint8_t* h_inliers, int* n_out) {
    DFVO_ARG_CHECK(p && n_out && (slot >= 0 && slot < DFVO_PIPELINE_SLOTS) && cap >= 0,
                   "dfvo_pipeline_get_keypoints: bad argument");
    TrackerBuffers& tb = p->tbs[slot];
    DFVO_HIP_CHECK(hipDeviceSynchronize());
    int info[3] = {0, 0, 0};
    DFVO_HIP_CHECK(hipMemcpy(info, tb.kp_info, sizeof(info), hipMemcpyDeviceToHost));
    *n_out = info[0];
    const int n = info[0] < cap ? info[0] : cap;
    if (n > 0 && h_kp_ref) DFVO_HIP_CHECK(hipMemcpy(h_kp_ref, tb.kp_ref, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
    if (n > 0 && h_kp_cur) DFVO_HIP_CHECK(hipMemcpy(h_kp_cur, tb.kp_cur, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
    if (n > 0 && h_inliers) DFVO_HIP_CHECK(hipMemcpy(h_inliers, tb.best_inliers, n, hipMemcpyDeviceToHost));
    return DFVO_OK;
}

int dfvo_pipeline_get_rng_state(dfvo_pipeline* p, uint32_t* h_state) {
    DFVO_ARG_CHECK(p && h_state, "dfvo_pipeline_get_rng_state: null argument");
    DFVO_HIP_CHECK(hipStreamSynchronize(p->s_trk));
    DFVO_HIP_CHECK(hipMemcpy(h_state, p->tbs[0].mt_state, 625 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return DFVO_OK;
}

int dfvo_pipeline_set_rng_state(dfvo_pipeline* p, const uint32_t* h_state) {
    DFVO_ARG_CHECK(p && h_state, "dfvo_pipeline_set_rng_state: null argument");
    DFVO_HIP_CHECK(hipStreamSynchronize(p->s_trk));
    DFVO_HIP_CHECK(hipMemcpy(p->tbs[0].mt_state, h_state, 625 * sizeof(uint32_t), hipMemcpyHostToDevice));
    return DFVO_OK;
}

int dfvo_pipeline_sync(dfvo_pipeline* p) {
    DFVO_ARG_CHECK(p, "null pipeline");
    DFVO_HIP_CHECK(hipStreamSynchronize(p->s_flow));
    for (int i = 0; i + 1 < p->flow_instances; ++i) DFVO_HIP_CHECK(hipStreamSynchronize(p->s_flow_x[i]));
    DFVO_HIP_CHECK(hipStreamSynchronize(p->s_depth));
    DFVO_HIP_CHECK(hipStreamSynchronize(p->s_trk));
    return DFVO_OK;
}

double dfvo_pipeline_net_flops(const dfvo_pipeline* p) { return p ? p->flow.flops_last + p->depth.flops_last : 0.0; }

}  // extern "C"
