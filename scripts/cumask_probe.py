"""Development probe: a pass with the caller's stream kept off some CUs (hipExtStreamCreateWithCUMask), so that the
geometry kernels of the side streams always find an empty CU beside the convolutions.
python scripts/cumask_probe.py [steps]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detection_3d_amd.config import get_cfg
from detection_3d_amd.detector import build_detection_model
from detection_3d_amd.synthetic import make_scene
from detection_3d_amd.voxelize import voxelize
from detection_3d_amd.serving import BuildingPipeline

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
cfg = get_cfg("4c_Fpn432")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev).eval()
scenes = [torch.from_numpy(make_scene(i, 500000)).to(dev) for i in range(4)]
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(off_bits):
    words = (ctypes.c_uint32 * 8)(*([0xffffffff] * 8))
    for b in off_bits:
        words[b // 32] &= ~(1 << (b % 32))
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


def run(tag, stream):
    s = cfg.SPARSE3D
    with torch.no_grad():
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    c, f = voxelize(scenes[i % 4], s.VOXEL_SCALE, s.VOXEL_FULL_SCALE)
                    model([c, f])
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
        print(f"{tag}: {dt / steps * 1e3:.3f} ms per building", flush=True)


run("default stream", None)
run("unmasked external stream", masked_stream([]))
run("bits 0-7 off", masked_stream(range(8)))
run("bits 0,32,..,224 off", masked_stream(range(0, 256, 32)))
run("bits 0-15 off", masked_stream(range(16)))
run("bits 0,16,..,240 off", masked_stream(range(0, 256, 16)))
run("bits 0-31 off", masked_stream(range(32)))
