"""Development probe: pooler (k_roi_sparse, bound by L2 probes) and the box head's first GEMM (matrix cores) over the
two halves of the proposals on two streams -- does the second half's pooling hide behind the first half's product?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize

dev = torch.device("cuda:0")
cfg = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).eval()
pcl = torch.from_numpy(make_scene(0, 500000)).to(dev)
with torch.no_grad():
    c, f = voxelize(pcl, 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)
    res, mid = model([c, f], return_intermediates=True)
    fe = model.roi_heads.box.feature_extractor
    conv = fe.conv3d[0]
    props = mid["proposals"]
    maps = mid["roi_features"]
    K = props.shape[0]
    w = conv.weight.view(conv.out_channels, -1).t().contiguous()
    print("proposals", K, flush=True)
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)

    def serial():
        pooled = fe.pooler.pool_metric(maps, props, fe.voxel_scale, channels_inner=True)
        Kp, ph, pw, C, pz = pooled.shape
        return torch.addmm(conv.bias, pooled.view(Kp * ph * pw, C * pz), w)

    def halves():
        h = K // 2
        pa = fe.pooler.pool_metric(maps, props[:h], fe.voxel_scale, channels_inner=True)
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            ya = torch.addmm(conv.bias, pa.view(h * pa.shape[1] * pa.shape[2], -1), w)
            ev2 = torch.cuda.Event(); ev2.record(side)
        pb = fe.pooler.pool_metric(maps, props[h:], fe.voxel_scale, channels_inner=True)
        yb = torch.addmm(conv.bias, pb.view((K - h) * pb.shape[1] * pb.shape[2], -1), w)
        main.wait_event(ev2)
        return ya, yb

    for name, fn in (("serial", serial), ("halves on two streams", halves), ("serial", serial), ("halves on two streams", halves)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
