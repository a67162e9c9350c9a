"""Library GEMMs of the detector tail with solutions picked ahead of time (torch's TunableOp, tuning OFF at run time).

The box head's two large fp32 products (roi_box_feature_extractors.py:47-117 -- the [1,1,pz] convolution as
a [K*cells, C*pz] x [C*pz, rep] GEMM and fc6) go to hipBLASLt through torch; its default heuristic picks solutions that
run at 80 / 98 TFLOP/s for these shapes, the best ones it holds reach 101 / 145.  `tuned/gemm_gfx950.csv` is the result
file of one tuning run on an MI355X (`scripts/tune_gemms.sh`; the validator lines pin torch, ROCm, hipBLASLt / rocBLAS
builds and the architecture -- on any mismatch TunableOp ignores the file and the default solutions run).  Shapes that
are not in the file (another proposal count, another config) take the default path; nothing is tuned at run time.
D3D_TUNED_GEMMS=0 switches it off.  Process-wide side effect: TunableOp is a torch-global switch, so while the table is
loaded EVERY GEMM of the host process consults it (shapes outside the table run the default solutions, as before); when
the table does not load, TunableOp is switched off again."""
import os
import shutil
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TABLE = os.path.join(HERE, "tuned", "gemm_gfx950.csv")
_state = {"done": False, "on": False}


def enable():
    """Idempotent; -> True when the table was handed to TunableOp."""
    if _state["done"]:
        return _state["on"]
    _state["done"] = True
    if os.environ.get("D3D_TUNED_GEMMS", "1") == "0" or not torch.cuda.is_available() or not os.path.exists(TABLE):
        return False
    if os.environ.get("PYTORCH_TUNABLEOP_ENABLED"):      # the user drives TunableOp: leave it alone
        return False
    try:
        import torch.cuda.tunable as tunable
        # TunableOp may rewrite its result file when the process exits: give it a private copy, never the tracked table
        workdir = tempfile.mkdtemp(prefix="d3d_gemm_")
        import atexit
        atexit.register(shutil.rmtree, workdir, True)
        work = os.path.join(workdir, "gemm.csv")
        shutil.copyfile(TABLE, work)
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.set_filename(work)
        _state["on"] = bool(tunable.read_file(work))
        if not _state["on"]:            # validators did not match this build: leave TunableOp as we found it (off)
            tunable.enable(False)
    except Exception as e:      # noqa: BLE001 -- an optional speed-up must never keep the model from being built
        import warnings
        warnings.warn("tuned GEMM table not used: %r (the library's default solutions run)" % (e,))
        try:
            tunable.enable(False)
        except Exception:       # noqa: BLE001
            pass
        _state["on"] = False
    return _state["on"]
