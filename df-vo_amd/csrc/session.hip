// Frame session under the drop-in classes (DeepModel / KeypointSampler / EssTracker mirrors): makes the reference's own
// synchronous call order (/root/reference/libs/dfvo.py:299-345 deep_model_inference, :121-262 tracking) fast WITHOUT changing
// what any call returns.
//
// The reference calls, per frame k:  forward_depth([img_k])  ->  forward_flow(cur = k, ref = k - 1)  ->  kp_selection  ->
// compute_pose_2d2d  ->  scale_recovery, each a blocking call with host numpy arrays in and out.  Executed literally that
// is five round trips with 3 x 1.4 MB of frame uploads, 9.3 MB of flow downloads and 9.3 MB of uploads again per pair,
// the depth net and the flow net back to back, and the 1.5 ms homography chain on the critical path (round 3-4: 104
// frames/s).  But everything the later calls need is known at the FIRST call: forward_depth(k) holds frame k, and frame
// k - 1 arrived with the previous call.  dfvo_session_push_frame therefore
//   * uploads the frame once (both nets read it on the device; the flow net carries frame k - 1's pyramids over),
//   * enqueues the depth net and the flow net of (k - 1, k) on their own streams, their outputs copied into pinned host
//     buffers (the arrays the mirrors return are views of those),
//   * enqueues, behind the flow net on the device, the keypoint selection and the RandomState-INDEPENDENT half of
//     compute_pose_2d2d (findHomography + refinement + GRIC-H) with the configuration the KeypointSampler / EssTracker
//     mirrors registered -- speculative, consumed only if the later calls turn out to ask for exactly that,
// and the later calls become "wait for an event, hand out the result".  Nothing is assumed: forward_flow checks that it is
// given the two frames that were pushed, kp_selection that the flow arrays it is handed are the session's (generation
// token + sampled contents, df-vo_amd/libs/deep_models/session.py), compute_pose_2d2d compares the keypoint arrays and
// the configuration byte for byte with what the device holds; any mismatch takes the plain host-array entry point.
// The RandomState-consuming half (shuffles, five-point RANSACs, recoverPose) needs np.random's state at the time of the
// compute_pose_2d2d call.  dfvo_session_pose_ahead (called by the DeepModel mirror from forward_flow, the first moment the host
// is back after the push) enqueues it under np.random's state AT THAT MOMENT; compute_pose_2d2d consumes the result only if
// np.random's state at its own call is still that state, word for word (and keypoints and configuration match) -- the
// reference's frame loop draws nothing in between (dfvo.py:324-333, 147-168) -- and otherwise waits for it, uploads the state
// it was called under and runs the plain entry point.  The scale recovery stays where the reference has it.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "capi_types.h"
#include "ops.h"

using namespace dfvo;

namespace {
constexpr int RING = 3;  // host buffer sets: a returned view stays valid until two further frames have been pushed
}

struct dfvo_session {
    dfvo_flownet* f = nullptr;
    dfvo_depthnet* d = nullptr;
    dfvo_tracker* t = nullptr;
    int H = 0, W = 0;
    hipStream_t s_copy = nullptr, s_pre = nullptr;
    uint8_t* d_img[2] = {nullptr, nullptr};
    long long gen = -1;          // frames pushed so far - 1 = generation of the newest frame
    bool carry_ok = false;       // the flow net's current-frame pyramids are those of frame `gen`
    bool have_flow = false;      // a flow pass of (gen - 1, gen) is enqueued / done
    bool have_kp = false, have_h = false;
    bool have_e = false;         // the RandomState-consuming half of compute_pose_2d2d is enqueued (dfvo_session_pose_ahead)
    uint32_t rng_ahead[625] = {};            // ... under this RandomState
    unsigned char ahead_cfg[sizeof(dfvo_pose2d2d_cfg)] = {};  // ... and this configuration (bytes)
    uint8_t* h_frame[RING] = {};  // pinned copies of the pushed frames: the upload's source, and what forward_flow's frames are compared with
    unsigned* h_ovf[RING] = {};   // [0] f16x3 range counter behind the flow net of the generation, [1] behind its depth net
    unsigned* d_ovf = nullptr;    // the device counter (conv_f16s_overflow_counter)
    unsigned ovf_seen[2] = {0, 0};  // its value up to which each of the two readers (flow stream, depth stream) has reported
    float* h_depth[RING] = {};
    float *h_fwd[RING] = {}, *h_bwd[RING] = {}, *h_diff[RING] = {};
    double *h_kp_ref[RING] = {}, *h_kp_cur[RING] = {};
    int* h_info[RING] = {};
    int kp_cap = 0;
    hipEvent_t e_img = nullptr, e_depth = nullptr, e_net = nullptr, e_flow = nullptr, e_kp = nullptr, e_pose = nullptr;
    dfvo_session_kp_cfg kp_cfg = {};
    dfvo_pose2d2d_cfg pose_cfg = {};
    // DFVO_SESSION_TRACE=1 (diagnostics): device times of the two nets of a push relative to the end of the frame upload
    bool trace = false;
    hipEvent_t t_img = nullptr, t_flow = nullptr, t_flow_copied = nullptr, t_depth = nullptr, t_kp = nullptr;
    int order = 0;  // DFVO_SESSION_ORDER (experiments): 0 flow net enqueued first | 1 depth net first | 2 depth net ON the flow
                    // net's stream, before it | 3 ... behind it
};

#define S_TRY(expr)                     \
    do {                                \
        int _rc = (expr);               \
        if (_rc != DFVO_OK) return _rc; \
    } while (0)

// Streams by dispatch pipe (stream_pool.hip): two busy streams on one of the command processor's four pipes slow each
// other's launches 2.5x, and which pipe a stream lands on follows the process's stream creation order -- with the nets'
// and the tracker's own streams (created whenever the host built those objects) the first session measured the depth net
// and the flow net SLOWER side by side than back to back (profiles/r5b_mirrors_first_run.txt: 7.2 vs 6.3 ms).  The session
// therefore measures a pool of candidates and re-homes every stream it drives: flow net | depth net + the speculative
// keypoint / homography stage | RandomState-ordered chain + its side streams | frame upload.  Streams the HOST handed to
// the nets / the tracker (own_stream false) are left alone.  Stream PRIORITIES were tried on top of the placement (round 5,
// profiles/r5n_session_stream_priorities.txt: flow net highest, depth net lowest, both): the time at which both nets are done
// does not move (4.7 ms either way, only which of the two finishes first), so every stream keeps the default priority; nor does
// it move when the depth net is held back until the flow net's Features stage is done (r5o_session_depth_after_features.txt): 4.7
// ms is the two nets' kernel time at one pair in flight (rocprofv3 timeline, r5p_mirrors_timeline.txt).
static int place_streams(dfvo_session* s) {
    StreamPool pool;
    if (pool.create(12) != DFVO_OK || pool.ngroups < 3) {
        pool.release();
        return DFVO_OK;  // (no measurement: creation-order streams, as before)
    }
    std::vector<int> order;
    for (int i = 0; i < pool.ngroups; ++i) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return pool.count(a) > pool.count(b); });
    const int g_trk = order[0], g_dep = order[1], g_flow = order[2], g_copy = pool.ngroups >= 4 ? order[3] : order[2];
    if (pool.count(g_trk) < 3 || pool.count(g_dep) < 2 || pool.count(g_flow) < (g_copy == g_flow ? 2 : 1) || pool.count(g_copy) < 1) {
        pool.release();
        return DFVO_OK;
    }
    DFVO_HIP_CHECK(hipDeviceSynchronize());
    auto rehome = [&](hipStream_t* slot, bool owned, int g) {
        if (!owned) return;
        hipStream_t n = pool.take(g);
        if (!n) return;
        if (*slot) (void)hipStreamDestroy(*slot);
        *slot = n;
    };
    rehome(&s->f->net.stream, s->f->net.own_stream, g_flow);
    rehome(&s->d->net.stream, s->d->net.own_stream, g_dep);
    rehome(&s->s_pre, true, g_dep);
    rehome(&s->s_copy, true, g_copy);
    if (s->t && s->t->own_stream && !s->t->tb.shared) {
        hipStream_t r0 = pool.take(g_trk), r1 = pool.take(g_trk);
        if (r0 && r1 && s->t->tb.rebind_streams(r0, r1) == DFVO_OK) {
            rehome(&s->t->stream, true, g_trk);
        } else {
            if (r0) (void)hipStreamDestroy(r0);
            if (r1) (void)hipStreamDestroy(r1);
        }
    }
    pool.release();
    return DFVO_OK;
}

extern "C" {

int dfvo_session_create(dfvo_flownet* f, dfvo_depthnet* d, dfvo_tracker* t, int img_h, int img_w, dfvo_session** out) {
    DFVO_ARG_CHECK(f && d && out && img_h > 0 && img_w > 0, "dfvo_session_create: bad argument");
    DFVO_ARG_CHECK(f->net.finalized && d->net.finalized, "dfvo_session_create: nets not finalized");
    DFVO_ARG_CHECK(f->net.imgH == img_h && f->net.imgW == img_w, "dfvo_session_create: flow net built for another image size");
    dfvo_session* s = new dfvo_session();
    s->f = f;
    s->d = d;
    s->t = t;
    s->H = img_h;
    s->W = img_w;
    const size_t px = (size_t)img_h * img_w, dpx = (size_t)d->net.H * d->net.W;
    bool ok = hipStreamCreateWithFlags(&s->s_copy, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&s->s_pre, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) ok = hipMalloc((void**)&s->d_img[i], px * 3) == hipSuccess;
    for (int i = 0; i < RING && ok; ++i) {
        ok = hipHostMalloc((void**)&s->h_depth[i], dpx * sizeof(float)) == hipSuccess &&
             hipHostMalloc((void**)&s->h_fwd[i], 2 * px * sizeof(float)) == hipSuccess &&
             hipHostMalloc((void**)&s->h_bwd[i], 2 * px * sizeof(float)) == hipSuccess &&
             hipHostMalloc((void**)&s->h_diff[i], px * sizeof(float)) == hipSuccess &&
             hipHostMalloc((void**)&s->h_info[i], 4 * sizeof(int)) == hipSuccess &&
             hipHostMalloc((void**)&s->h_frame[i], px * 3) == hipSuccess &&
             hipHostMalloc((void**)&s->h_ovf[i], 2 * sizeof(unsigned)) == hipSuccess;
        if (ok) s->h_ovf[i][0] = s->h_ovf[i][1] = 0;
    }
    if (ok) {
        unsigned long long seen = 0;
        s->d_ovf = conv_f16s_overflow_counter();
        ok = s->d_ovf != nullptr && conv_f16s_overflow_count(&seen, 0) == DFVO_OK;
        unsigned dev = 0;
        ok = ok && hipMemcpy(&dev, s->d_ovf, sizeof(dev), hipMemcpyDeviceToHost) == hipSuccess;
        s->ovf_seen[0] = s->ovf_seen[1] = dev;  // (events before the session -- other nets of the process -- are not this session's)
    }
    hipEvent_t* evs[6] = {&s->e_img, &s->e_depth, &s->e_net, &s->e_flow, &s->e_kp, &s->e_pose};
    for (int i = 0; i < 6 && ok; ++i) ok = hipEventCreateWithFlags(evs[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        dfvo_session_destroy(s);
        dfvo::set_last_error("dfvo_session_create: allocation failed");
        return DFVO_ERR_HIP;
    }
    s->trace = getenv("DFVO_SESSION_TRACE") != nullptr;
    s->order = getenv("DFVO_SESSION_ORDER") ? atoi(getenv("DFVO_SESSION_ORDER")) : 0;
    if (s->trace) {
        hipEvent_t* tv[5] = {&s->t_img, &s->t_flow, &s->t_flow_copied, &s->t_depth, &s->t_kp};
        for (hipEvent_t* e : tv) (void)hipEventCreate(e);
    }
    if (place_streams(s) != DFVO_OK) {
        dfvo_session_destroy(s);
        return DFVO_ERR_HIP;
    }
    *out = s;
    return DFVO_OK;
}

void dfvo_session_destroy(dfvo_session* s) {
    if (!s) return;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i)
        if (s->d_img[i]) (void)hipFree(s->d_img[i]);
    for (int i = 0; i < RING; ++i) {
        void* hp[9] = {s->h_depth[i], s->h_fwd[i], s->h_bwd[i], s->h_diff[i], s->h_kp_ref[i], s->h_kp_cur[i], s->h_info[i],
                       s->h_frame[i], s->h_ovf[i]};
        for (void* q : hp)
            if (q) (void)hipHostFree(q);
    }
    hipEvent_t evs[6] = {s->e_img, s->e_depth, s->e_net, s->e_flow, s->e_kp, s->e_pose};
    for (hipEvent_t e : evs)
        if (e) (void)hipEventDestroy(e);
    if (s->s_copy) (void)hipStreamDestroy(s->s_copy);
    if (s->s_pre) (void)hipStreamDestroy(s->s_pre);
    delete s;
}

int dfvo_session_reset(dfvo_session* s) {
    DFVO_ARG_CHECK(s, "dfvo_session_reset: null session");
    DFVO_HIP_CHECK(hipDeviceSynchronize());
    s->carry_ok = s->have_flow = s->have_kp = s->have_h = s->have_e = false;
    s->gen = -1;
    return DFVO_OK;
}

// another caller ran the flow net (LiteFlow.inference_flow ...): its pyramids are no longer the held frame's
int dfvo_session_invalidate_carry(dfvo_session* s) {
    DFVO_ARG_CHECK(s, "dfvo_session_invalidate_carry: null session");
    s->carry_ok = false;
    return DFVO_OK;
}

// a plain solver entry point is about to use the tracker's buffers: the speculative stage must not be running on them
int dfvo_session_quiesce(dfvo_session* s) {
    DFVO_ARG_CHECK(s, "dfvo_session_quiesce: null session");
    DFVO_HIP_CHECK(hipStreamSynchronize(s->s_pre));
    if (s->have_e && s->t) DFVO_HIP_CHECK(hipStreamSynchronize(s->t->stream));
    s->have_h = s->have_e = false;  // their buffers are about to be overwritten: compute_pose_2d2d must restage
    return DFVO_OK;
}

static int ensure_kp_host(dfvo_session* s, int cap) {
    if (cap <= s->kp_cap) return DFVO_OK;
    DFVO_HIP_CHECK(hipDeviceSynchronize());  // (a copy into the old buffers may still be in flight)
    for (int i = 0; i < RING; ++i) {
        if (s->h_kp_ref[i]) (void)hipHostFree(s->h_kp_ref[i]);
        if (s->h_kp_cur[i]) (void)hipHostFree(s->h_kp_cur[i]);
        s->h_kp_ref[i] = s->h_kp_cur[i] = nullptr;
        DFVO_HIP_CHECK(hipHostMalloc((void**)&s->h_kp_ref[i], sizeof(double) * 2 * cap));
        DFVO_HIP_CHECK(hipHostMalloc((void**)&s->h_kp_cur[i], sizeof(double) * 2 * cap));
    }
    s->kp_cap = cap;
    return DFVO_OK;
}

int dfvo_session_push_frame(dfvo_session* s, const uint8_t* h_img, const dfvo_session_kp_cfg* kp,
                            const dfvo_pose2d2d_cfg* pose, int flags, long long* generation) {
    DFVO_ARG_CHECK(s && h_img, "dfvo_session_push_frame: bad argument");
    const bool want_flow = !(flags & DFVO_PUSH_NO_FLOW);
    FlowNet& fn = s->f->net;
    DepthNet& dn = s->d->net;
    const size_t px = (size_t)s->H * s->W, dpx = (size_t)dn.H * dn.W;
    const long long g = s->gen + 1;
    const int slot = (int)(g % RING);
    uint8_t* img = s->d_img[g & 1];
    // (the device buffer of frame g - 2 is reused: the nets that read it were waited for by the calls of pair g - 1, or are
    // ordered before this copy by the events below)
    DFVO_HIP_CHECK(hipStreamWaitEvent(s->s_copy, s->e_depth, 0));
    DFVO_HIP_CHECK(hipStreamWaitEvent(s->s_copy, s->e_net, 0));
    // the frame goes through the session's own pinned copy (slot g % RING: the upload that read it last was that of frame
    // g - RING, waited for long ago): the caller's buffer is its own again when this returns, the copy is asynchronous, and
    // dfvo_session_frame hands the copy out for the byte-for-byte comparison forward_flow makes (no sampling)
    memcpy(s->h_frame[slot], h_img, px * 3);
    DFVO_HIP_CHECK(hipMemcpyAsync(img, s->h_frame[slot], px * 3, hipMemcpyHostToDevice, s->s_copy));
    DFVO_HIP_CHECK(hipEventRecord(s->e_img, s->s_copy));
    if (s->trace) DFVO_HIP_CHECK(hipEventRecord(s->t_img, s->s_copy));
    s->have_flow = s->have_kp = s->have_h = s->have_e = false;
    // ---- flow net of (g - 1, g)
    auto enqueue_flow = [&]() -> int {
        if (g >= 1 && want_flow) {
            DFVO_HIP_CHECK(hipStreamWaitEvent(fn.stream, s->e_img, 0));
            if (s->carry_ok)
                S_TRY(fn.forward(nullptr, img, fn.out_fwd.p, fn.out_bwd.p, fn.out_diff.p, &fn));
            else
                S_TRY(fn.forward(s->d_img[(g - 1) & 1], img, fn.out_fwd.p, fn.out_bwd.p, fn.out_diff.p));
            DFVO_HIP_CHECK(hipEventRecord(s->e_net, fn.stream));
            if (s->trace) DFVO_HIP_CHECK(hipEventRecord(s->t_flow, fn.stream));
            DFVO_HIP_CHECK(hipMemcpyAsync(&s->h_ovf[slot][0], s->d_ovf, sizeof(unsigned), hipMemcpyDeviceToHost, fn.stream));
            DFVO_HIP_CHECK(hipMemcpyAsync(s->h_fwd[slot], fn.out_fwd.p, 2 * px * sizeof(float), hipMemcpyDeviceToHost, fn.stream));
            DFVO_HIP_CHECK(hipMemcpyAsync(s->h_bwd[slot], fn.out_bwd.p, 2 * px * sizeof(float), hipMemcpyDeviceToHost, fn.stream));
            DFVO_HIP_CHECK(hipMemcpyAsync(s->h_diff[slot], fn.out_diff.p, px * sizeof(float), hipMemcpyDeviceToHost, fn.stream));
            DFVO_HIP_CHECK(hipEventRecord(s->e_flow, fn.stream));
            if (s->trace) DFVO_HIP_CHECK(hipEventRecord(s->t_flow_copied, fn.stream));
            s->have_flow = true;
        } else {
            // first frame (or a push without the flow pass): its pyramids come into being with the next pair (both frames
            // through Features)
            DFVO_HIP_CHECK(hipEventRecord(s->e_net, fn.stream));
        }
        s->carry_ok = g >= 1 && want_flow;
        return DFVO_OK;
    };
    // ---- depth net of frame g (Pillow-exact LANCZOS resize of the full frame on the device, then the net)
    auto enqueue_depth = [&]() -> int {
        if (s->d->resize.H != s->H || s->d->resize.W != s->W || s->d->resize.oh != dn.H || s->d->resize.ow != dn.W)
            S_TRY(s->d->resize.init(s->H, s->W, dn.H, dn.W));
        DFVO_HIP_CHECK(hipStreamWaitEvent(dn.stream, s->e_img, 0));
        S_TRY(s->d->resize.enqueue(img, (uint8_t*)dn.u8_in.p, dn.stream));
        S_TRY(dn.forward((const uint8_t*)dn.u8_in.p, dn.depth.p));
        DFVO_HIP_CHECK(hipMemcpyAsync(s->h_depth[slot], dn.depth.p, dpx * sizeof(float), hipMemcpyDeviceToHost, dn.stream));
        DFVO_HIP_CHECK(hipMemcpyAsync(&s->h_ovf[slot][1], s->d_ovf, sizeof(unsigned), hipMemcpyDeviceToHost, dn.stream));
        DFVO_HIP_CHECK(hipEventRecord(s->e_depth, dn.stream));
        if (s->trace) DFVO_HIP_CHECK(hipEventRecord(s->t_depth, dn.stream));
        return DFVO_OK;
    };
    if (s->order == 1) {
        S_TRY(enqueue_depth());
        S_TRY(enqueue_flow());
    } else if (s->order == 2 || s->order == 3) {
        hipStream_t keep = dn.stream;
        dn.stream = fn.stream;  // (a captured graph may be launched into any stream)
        int rc = DFVO_OK;
        if (s->order == 2) rc = enqueue_depth();
        if (rc == DFVO_OK) rc = enqueue_flow();
        if (rc == DFVO_OK && s->order == 3) rc = enqueue_depth();
        dn.stream = keep;
        S_TRY(rc);
    } else {
        S_TRY(enqueue_flow());
        S_TRY(enqueue_depth());
    }
    // ---- speculative keypoint selection + RandomState-independent half of compute_pose_2d2d, behind the flow net
    if (g >= 1 && want_flow && kp && s->t) {
        TrackerBuffers& tb = s->t->tb;
        DFVO_HIP_CHECK(hipStreamWaitEvent(s->s_pre, s->e_net, 0));
        DFVO_HIP_CHECK(hipStreamWaitEvent(s->s_pre, s->e_pose, 0));  // (an early pose half nobody consumed may still be running)
        S_TRY(enqueue_local_bestn(tb, fn.out_fwd.p, fn.out_diff.p, s->H, s->W, kp->num_row, kp->num_col, kp->num_bestN, kp->thre,
                                  s->s_pre, kp->score_method));
        S_TRY(ensure_kp_host(s, tb.kp_cap));
        DFVO_HIP_CHECK(hipMemcpyAsync(s->h_info[slot], tb.kp_info, 3 * sizeof(int), hipMemcpyDeviceToHost, s->s_pre));
        DFVO_HIP_CHECK(hipMemcpyAsync(s->h_kp_ref[slot], tb.kp_ref, sizeof(double) * 2 * tb.kp_cap, hipMemcpyDeviceToHost, s->s_pre));
        DFVO_HIP_CHECK(hipMemcpyAsync(s->h_kp_cur[slot], tb.kp_cur, sizeof(double) * 2 * tb.kp_cap, hipMemcpyDeviceToHost, s->s_pre));
        DFVO_HIP_CHECK(hipEventRecord(s->e_kp, s->s_pre));
        if (s->trace) DFVO_HIP_CHECK(hipEventRecord(s->t_kp, s->s_pre));
        s->kp_cfg = *kp;
        s->have_kp = true;
        if (pose) {
            PoseConfig pc;
            S_TRY(dfvo_pose_config_from(pose, &pc));
            S_TRY(enqueue_pose_h_part(tb, tb.kp_cap, pc, s->s_pre));  // keypoint count read on the device
            s->pose_cfg = *pose;
            s->have_h = true;
        }
    }
    s->gen = g;
    if (generation) *generation = g;
    return DFVO_OK;
}

// The f16x3 / f16 split has f16's range: an activation beyond +-65504 becomes inf in the layer that splits it (conv_win_f16s.h).
// The device counter of such events rides behind each net's output copy; a net that raised it fails its call here instead of
// handing out inf / NaN silently (the exact-fp32 packing never counts).  The counter is process-wide: the two nets of a push run
// side by side, so an event in either fails whichever of forward_depth / forward_flow reads the counter after it -- possibly
// both; each reader keeps its own high-water mark (its reads are ordered on its stream).
static int range_check(dfvo_session* s, int reader, unsigned now, const char* which) {
    unsigned& seen = s->ovf_seen[reader];
    if (now < seen) seen = 0;  // (someone reset the counter: dfvo_f16s_overflow_count(.., 1))
    if (now == seen) return DFVO_OK;
    const unsigned n = now - seen;
    seen = now;
    dfvo::set_last_error(std::string("f16 split out of range: ") + std::to_string(n) + " activation group(s) beyond +-65504 up to the " +
                         which + " net of this frame -- its output holds inf / NaN; pack the nets in exact fp32 "
                         "(DFVO_CONV_PRECISION=fp32 or dfvo_hip.conv_precision: fp32)");
    return DFVO_ERR_RANGE;
}

int dfvo_session_depth(dfvo_session* s, long long generation, const float** h_depth) {
    DFVO_ARG_CHECK(s && h_depth, "dfvo_session_depth: bad argument");
    DFVO_ARG_CHECK(generation == s->gen && s->gen >= 0, "dfvo_session_depth: not the newest pushed frame");
    DFVO_HIP_CHECK(hipEventSynchronize(s->e_depth));
    *h_depth = s->h_depth[s->gen % RING];
    return range_check(s, 1, s->h_ovf[s->gen % RING][1], "depth");
}

// the session's pinned copy of frame `generation` (the newest or the one before): uint8 [img_h, img_w, 3]
int dfvo_session_frame(dfvo_session* s, long long generation, const uint8_t** h_frame) {
    DFVO_ARG_CHECK(s && h_frame, "dfvo_session_frame: bad argument");
    DFVO_ARG_CHECK(generation >= 0 && generation <= s->gen && generation >= s->gen - 1, "dfvo_session_frame: frame no longer held");
    *h_frame = s->h_frame[generation % RING];
    return DFVO_OK;
}

// The host buffers of `generation`'s ring slot (depth, fwd, bwd, diff) are still referenced by the caller while the slot is
// about to be reused (or the session destroyed): the session allocates itself fresh ones and hands the old ones over --
// h_old4 = {depth, fwd, bwd, diff}, to be released with dfvo_host_free when the last reference is gone.
int dfvo_session_detach_slot(dfvo_session* s, long long generation, void** h_old4) {
    DFVO_ARG_CHECK(s && h_old4 && generation >= 0, "dfvo_session_detach_slot: bad argument");
    const int i = (int)(generation % RING);
    const size_t px = (size_t)s->H * s->W, dpx = (size_t)s->d->net.H * s->d->net.W;
    DFVO_HIP_CHECK(hipDeviceSynchronize());  // (no copy into the old buffers is in flight any more)
    float *nd = nullptr, *nf = nullptr, *nb = nullptr, *nx = nullptr;
    const bool ok = hipHostMalloc((void**)&nd, dpx * sizeof(float)) == hipSuccess && hipHostMalloc((void**)&nf, 2 * px * sizeof(float)) == hipSuccess &&
                    hipHostMalloc((void**)&nb, 2 * px * sizeof(float)) == hipSuccess && hipHostMalloc((void**)&nx, px * sizeof(float)) == hipSuccess;
    if (!ok) {
        void* fresh[4] = {nd, nf, nb, nx};
        for (void* q : fresh)
            if (q) (void)hipHostFree(q);
        dfvo::set_last_error("dfvo_session_detach_slot: pinned allocation failed");
        return DFVO_ERR_HIP;
    }
    h_old4[0] = s->h_depth[i];
    h_old4[1] = s->h_fwd[i];
    h_old4[2] = s->h_bwd[i];
    h_old4[3] = s->h_diff[i];
    s->h_depth[i] = nd;
    s->h_fwd[i] = nf;
    s->h_bwd[i] = nb;
    s->h_diff[i] = nx;
    return DFVO_OK;
}

int dfvo_host_free(void* h_pinned) {
    if (h_pinned) DFVO_HIP_CHECK(hipHostFree(h_pinned));
    return DFVO_OK;
}

int dfvo_session_flow(dfvo_session* s, long long generation, const float** h_fwd, const float** h_bwd, const float** h_diff) {
    DFVO_ARG_CHECK(s && h_fwd && h_bwd && h_diff, "dfvo_session_flow: bad argument");
    DFVO_ARG_CHECK(generation == s->gen && s->have_flow, "dfvo_session_flow: no flow pass of that generation is held");
    DFVO_HIP_CHECK(hipEventSynchronize(s->e_flow));
    if (s->trace) {
        float a = 0, b = 0, c = 0, d = 0;
        (void)hipEventSynchronize(s->t_depth);
        (void)hipEventElapsedTime(&a, s->t_img, s->t_flow);
        (void)hipEventElapsedTime(&b, s->t_img, s->t_flow_copied);
        (void)hipEventElapsedTime(&c, s->t_img, s->t_depth);
        if (s->have_kp && hipEventSynchronize(s->t_kp) == hipSuccess) (void)hipEventElapsedTime(&d, s->t_img, s->t_kp);
        fprintf(stderr, "dfvo session trace gen %lld: after the upload (ms): flow net %.3f  + copies %.3f | depth net + copy %.3f | keypoints on the host %.3f\n",
                s->gen, a, b, c, d);
    }
    const int slot = (int)(s->gen % RING);
    *h_fwd = s->h_fwd[slot];
    *h_bwd = s->h_bwd[slot];
    *h_diff = s->h_diff[slot];
    return range_check(s, 0, s->h_ovf[slot][0], "flow");
}

int dfvo_session_keypoints(dfvo_session* s, long long generation, const dfvo_session_kp_cfg* kp, const double** h_kp_ref,
                           const double** h_kp_cur, int* n, int* good_kp_found) {
    DFVO_ARG_CHECK(s && kp && h_kp_ref && h_kp_cur && n && good_kp_found, "dfvo_session_keypoints: bad argument");
    DFVO_ARG_CHECK(generation == s->gen && s->have_kp, "dfvo_session_keypoints: no keypoint selection of that generation is held");
    DFVO_ARG_CHECK(memcmp(kp, &s->kp_cfg, sizeof(*kp)) == 0, "dfvo_session_keypoints: selection was run with another configuration");
    DFVO_HIP_CHECK(hipEventSynchronize(s->e_kp));
    const int slot = (int)(s->gen % RING);
    *n = s->h_info[slot][0];
    *good_kp_found = s->h_info[slot][1];
    *h_kp_ref = s->h_kp_ref[slot];
    *h_kp_cur = s->h_kp_cur[slot];
    return DFVO_OK;
}

// does (kp arrays, configuration) ask for what the homography half of this generation was enqueued with?
static bool h_half_matches(dfvo_session* s, const double* h_kp_ref, const double* h_kp_cur, int n, const dfvo_pose2d2d_cfg* cfg) {
    const int slot = (int)(s->gen % RING);
    // the homography half reads: the keypoints, validity method / threshold, K^-T, K^-1 (enqueue_pose_h_part)
    return s->h_info[slot][1] != 0 && s->h_info[slot][0] == n && n <= s->kp_cap &&
           memcmp(h_kp_ref, s->h_kp_ref[slot], sizeof(double) * 2 * n) == 0 &&
           memcmp(h_kp_cur, s->h_kp_cur[slot], sizeof(double) * 2 * n) == 0 &&
           cfg->validity_method == s->pose_cfg.validity_method && cfg->validity_thre == s->pose_cfg.validity_thre &&
           memcmp(cfg->KinvT, s->pose_cfg.KinvT, sizeof(cfg->KinvT)) == 0 && memcmp(cfg->Kinv, s->pose_cfg.Kinv, sizeof(cfg->Kinv)) == 0;
}

// The RandomState-consuming half of compute_pose_2d2d, enqueued as soon as the keypoints of this generation exist, under the
// RandomState the caller has NOW (h_rng625: numpy's 624 key words + position).  *enqueued = 1 when it was (the keypoint
// selection found keypoints and the homography half runs with this configuration); dfvo_session_pose_2d2d hands its result
// out if it is then called under the same RandomState, keypoints and configuration, and discards it otherwise.
int dfvo_session_pose_ahead(dfvo_session* s, long long generation, const uint32_t* h_rng625, const dfvo_pose2d2d_cfg* cfg,
                            int* enqueued) {
    DFVO_ARG_CHECK(s && h_rng625 && cfg && enqueued, "dfvo_session_pose_ahead: bad argument");
    *enqueued = 0;
    if (!(s->t && generation == s->gen && s->gen >= 1 && s->have_kp && s->have_h && !s->have_e)) return DFVO_OK;
    DFVO_HIP_CHECK(hipEventSynchronize(s->e_kp));
    const int slot = (int)(s->gen % RING);
    const int n = s->h_info[slot][0];
    if (!h_half_matches(s, s->h_kp_ref[slot], s->h_kp_cur[slot], n, cfg)) return DFVO_OK;  // (no keypoints / another configuration)
    PoseConfig pc;
    S_TRY(dfvo_pose_config_from(cfg, &pc));
    S_TRY(dfvo_tracker_set_rng_state(s->t, h_rng625));
    S_TRY(enqueue_pose_e_part(s->t->tb, n, pc, s->t->stream, nullptr));  // waits for the homography half (tb.ev_h) on the device
    DFVO_HIP_CHECK(hipEventRecord(s->e_pose, s->t->stream));
    memcpy(s->rng_ahead, h_rng625, sizeof(s->rng_ahead));
    memcpy(s->ahead_cfg, cfg, sizeof(s->ahead_cfg));
    s->have_e = true;
    *enqueued = 1;
    return DFVO_OK;
}

// EssTracker.compute_pose_2d2d through the session, under the RandomState h_rng625 (np.random's state at the call).
// *used_resident: 2 = the whole call had been enqueued ahead under exactly this RandomState, these keypoints and this
// configuration (dfvo_session_pose_ahead): its result is handed out; 1 = the keypoints handed in are, byte for byte, the ones
// the device selected for this generation and the homography half was enqueued with this very configuration: only the
// RandomState-consuming half runs now; 0 = the plain entry point ran.
int dfvo_session_pose_2d2d(dfvo_session* s, const double* h_kp_ref, const double* h_kp_cur, int n,
                           const dfvo_pose2d2d_cfg* cfg, dfvo_pose2d2d_out* out, uint8_t* h_inliers, const uint32_t* h_rng625,
                           int* used_resident) {
    DFVO_ARG_CHECK(s && s->t && h_kp_ref && h_kp_cur && cfg && out && h_inliers && h_rng625 && n >= 0,
                   "dfvo_session_pose_2d2d: bad argument");
    bool resident = s->have_kp && s->have_h && s->gen >= 1;
    if (resident) {
        DFVO_HIP_CHECK(hipEventSynchronize(s->e_kp));
        resident = h_half_matches(s, h_kp_ref, h_kp_cur, n, cfg);
    }
    const bool had_ahead = s->have_e;
    const bool ahead = had_ahead && resident && memcmp(cfg, s->ahead_cfg, sizeof(s->ahead_cfg)) == 0 &&
                       memcmp(h_rng625, s->rng_ahead, sizeof(s->rng_ahead)) == 0;
    s->have_h = s->have_e = false;  // both halves are consumed (or abandoned) by this call
    if (used_resident) *used_resident = ahead ? 2 : (resident && !had_ahead) ? 1 : 0;
    if (ahead) return dfvo_pose_fetch(s->t, n, cfg, out, h_inliers);
    if (had_ahead) {
        // enqueued under another RandomState / configuration / keypoint set: let it finish, discard it; the plain entry point
        // restages everything it needs
        DFVO_HIP_CHECK(hipStreamSynchronize(s->t->stream));
        resident = false;
    }
    S_TRY(dfvo_tracker_set_rng_state(s->t, h_rng625));
    if (!resident) {
        // the speculative half may still be running on the tracker's buffers: let it finish before they are restaged
        DFVO_HIP_CHECK(hipStreamSynchronize(s->s_pre));
        return dfvo_compute_pose_2d2d(s->t, h_kp_ref, h_kp_cur, n, cfg, out, h_inliers);
    }
    PoseConfig pc;
    S_TRY(dfvo_pose_config_from(cfg, &pc));
    S_TRY(enqueue_pose_e_part(s->t->tb, n, pc, s->t->stream, nullptr));  // waits for the homography half (tb.ev_h) on the device
    return dfvo_pose_fetch(s->t, n, cfg, out, h_inliers);
}

}  // extern "C"
