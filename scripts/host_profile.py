"""Host side of one building's pass: cProfile over 20 passes (sorted by own time) -- which Python / ctypes calls the
launch thread spends its time in.  python scripts/host_profile.py [points]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize

n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
dev = torch.device("cuda:0")
cfg = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).eval()
clouds = [torch.from_numpy(make_scene(i, n_points)).to(dev) for i in range(4)]


def run(n):
    with torch.no_grad():
        for i in range(n):
            model(list(voxelize(clouds[i % 4], 50, cfg.SPARSE3D.VOXEL_FULL_SCALE)))
    torch.cuda.synchronize()


run(8)
pr = cProfile.Profile()
pr.enable()
run(20)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
# the detector tail alone: cumulative time of its Python functions and of the tensor-library calls under them
st.sort_stats("cumtime").print_stats(r"detector\.py|box_ops\.py|roi_align_rotated_3d\.py|topk|sigmoid|addmm|linear|softmax|'sort'|'item'|synchronize|record", 40)
