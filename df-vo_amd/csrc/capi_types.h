// The opaque handle types of include/dfvo_hip.h, shared by the translation units of the extern "C" surface
// (capi.hip: nets and operators, capi_tracker.hip: solvers, session.hip: the frame session under the drop-in classes).
#pragma once
#include "../../include/dfvo_hip.h"
#include "nets.h"
#include "resize_lanczos.h"
#include "solver.h"
#include "tracker.h"

struct dfvo_flownet {
    dfvo::FlowNet net;
};
struct dfvo_depthnet {
    dfvo::DepthNet net;
    dfvo::LanczosResizer resize;  // dfvo_depthnet_forward_image_host / the session: tables for the last image size seen
    uint8_t* img_full = nullptr;
    size_t img_full_bytes = 0;
};
struct dfvo_tracker {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    dfvo::TrackerBuffers tb;
    dfvo::RigidKpBuffers rigid;
    dfvo::BestNBuffers bestn;
    dfvo::RansacWorkspace& ws = tb.ws_e;
    float *d_flow = nullptr, *d_diff = nullptr;
    double* d_depth = nullptr;
    size_t flow_cap = 0, depth_cap = 0;
    double* d_small = nullptr;  // 64 doubles
    double *d_x1 = nullptr, *d_x2 = nullptr, *d_X4 = nullptr;
    int tri_cap = 0;
    dfvo::PnpBuffers pnp;
};


// internal helpers of capi_tracker.hip used by session.hip
int dfvo_pose_config_from(const dfvo_pose2d2d_cfg* cfg, dfvo::PoseConfig* pc);
int dfvo_pose_fetch(dfvo_tracker* t, int n, const dfvo_pose2d2d_cfg* cfg, dfvo_pose2d2d_out* out, uint8_t* h_inliers);
