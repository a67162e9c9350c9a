"""The stable radix sort behind plan row order and point lists (d3d_sort_pairs) against numpy's stable sort: bit-exact
permutations (integer work), sizes from one element to several tiles with ragged ends, few distinct keys (long runs of
equal digits: stability), ascending / descending, every digit layout (8-, 9- and 10-bit digits, 1 to 4 passes), and the
order it gives a rulebook's rows."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sort(dev, keys, vals, bits, desc, want_keys=True):
    from detection_3d_amd._lib import check, lib, ptr, stream_of
    n = keys.shape[0]
    k = torch.from_numpy(keys.view(np.int32)).to(dev)
    v = torch.from_numpy(vals).to(dev)
    ko = torch.empty_like(k)
    vo = torch.empty_like(v)
    nb = lib().d3d_sort_scratch_bytes(n, bits)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    check(lib().d3d_sort_pairs(ptr(k), ptr(v), n, bits, int(desc), ptr(ko) if want_keys else None, ptr(vo),
                               ptr(scratch), nb, stream_of()))
    torch.cuda.synchronize()
    return ko.cpu().numpy().view(np.uint32), vo.cpu().numpy()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 2047, 2048, 2049, 5000, 70001, 300000])
@pytest.mark.parametrize("bits,distinct", [(32, None), (27, None), (20, None), (19, None), (13, 40), (10, None), (9, None),
                                           (8, 3), (1, 2)])
@pytest.mark.parametrize("desc", [False, True])
def test_sort_pairs_is_the_stable_sort(dev, n, bits, distinct, desc):
    rng = np.random.RandomState(n * 31 + bits)
    hi = 1 << bits
    if distinct:
        pool = rng.randint(0, hi, distinct, dtype=np.int64)
        keys = pool[rng.randint(0, distinct, n)].astype(np.uint32)
    else:
        keys = rng.randint(0, hi, n, dtype=np.int64).astype(np.uint32)
    vals = rng.permutation(n).astype(np.int32)
    ko, vo = _sort(dev, keys, vals, bits, desc)
    order = np.argsort(-keys.astype(np.int64) if desc else keys.astype(np.int64), kind="stable")
    assert np.array_equal(ko, keys[order])            # bit-exact
    assert np.array_equal(vo, vals[order])
    _, vo2 = _sort(dev, keys, vals, bits, desc, want_keys=False)
    assert np.array_equal(vo2, vals[order])


@pytest.mark.parametrize("K", [8, 27, 32])
def test_rulebook_row_order(dev, K):
    """finalize_plan sorts the rows' keys (popcount << K | offset mask) by their low K bits, descending: rows of one mask
    end up contiguous and in their original (site) order, whatever sits above the mask in the key."""
    n = 50000
    rng = np.random.RandomState(K)
    masks = rng.randint(1, 1 << min(K, 31), 300, dtype=np.int64).astype(np.uint32)
    m = masks[rng.randint(0, 300, n)]
    pop = np.array([bin(int(x)).count("1") for x in m], np.uint32)
    keys = (((pop << np.uint32(K)) | m) if K <= 27 else m).astype(np.uint32)
    vals = np.arange(n, dtype=np.int32)
    _, vo = _sort(dev, keys, vals, min(K, 32), True, want_keys=False)
    order = np.argsort(-m.astype(np.int64), kind="stable")
    assert np.array_equal(vo, vals[order])                       # bit-exact
    sm = m[vo]
    assert np.flatnonzero(np.diff(sm.astype(np.int64)) != 0).size + 1 == np.unique(m).size
    for mask in masks[:20]:
        assert (np.diff(vo[sm == mask]) > 0).all()
