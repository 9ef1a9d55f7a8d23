"""The keys of the reference's options/examples/default_configuration.yml that the tracking hot path reads, as the
attribute-style mapping libs/dfvo.py passes around (the reference builds it with EasyDict from the YAML file).
Used by bench.py --surface mirrors and by hosts that drive the mirror classes without the reference's YAML loader."""


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def default_configuration(height, width, flow_weight_path, depth_model_dir):
    """default_configuration.yml: hybrid tracking, local_bestN 2000 keypoints in 10 x 10 cells, GRIC validity, simple
    depth-ratio scale recovery, PnP fallback (5 x 100 iterations), monodepth2 + LiteFlowNet with forward-backward flow"""
    it = Cfg(enable=False, kp_src="kp_depth", score_method="opt_flow")
    return Cfg(
        dataset="kitti_odom", seed=4869, tracking_method="hybrid",
        image=Cfg(height=height, width=width),
        crop=Cfg(depth_crop=[[0.3, 1], [0, 1]], flow_crop=[[0, 1], [0, 1]]),
        depth=Cfg(depth_src=None, min_depth=0.0, max_depth=50.0,
                  deep_depth=Cfg(network="monodepth2", pretrained_model=depth_model_dir)),
        deep_flow=Cfg(network="liteflow", flow_net_weight=flow_weight_path, forward_backward=True),
        deep_pose=Cfg(enable=False),
        online_finetune=Cfg(enable=False, flow=Cfg(enable=True), depth=Cfg(enable=False)),
        kp_selection=Cfg(local_bestN=Cfg(enable=True, num_bestN=2000, num_row=10, num_col=10, score_method="flow", thre=0.1),
                         bestN=Cfg(enable=False, num_bestN=2000), sampled_kp=Cfg(enable=False, num_kp=2000),
                         rigid_flow_kp=Cfg(enable=False, num_bestN=2000, num_row=10, num_col=10, score_method="opt_flow",
                                           rigid_flow_thre=5, optical_flow_thre=0.1),
                         depth_consistency=Cfg(enable=False, thre=0.05)),
        e_tracker=Cfg(ransac=Cfg(reproj_thre=0.2, repeat=5), validity=Cfg(method="GRIC", thre=None), kp_src="kp_best",
                      iterative_kp=Cfg(it)),
        scale_recovery=Cfg(method="simple", kp_src="kp_best", iterative_kp=Cfg(it),
                           ransac=Cfg(method="depth_ratio", min_samples=3, max_trials=100, stop_prob=0.99, thre=0.1)),
        pnp_tracker=Cfg(ransac=Cfg(iter=100, reproj_thre=1.0, repeat=5), kp_src="kp_best", iterative_kp=Cfg(it)))
