"""-m gpu: the index build's radix sort (scn_sort_pairs, csrc/scn_sort.hip) against numpy's stable sort -- bit-exact.
Sizes straddle the one-workgroup form (<= 4096 pairs), wave and workgroup boundaries, and the sizes of the path (3 k ...
600 k rows); key widths cover one, two and three passes and uneven digit widths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sort(gpu, keys, vals, bits):
    from sparse_rcnn_amd import _lib as L
    lib = L.lib()
    n = len(keys)
    k = torch.from_numpy(keys.view(np.int32)).to(gpu)
    v = torch.from_numpy(vals).to(gpu) if vals is not None else None
    ko = torch.full((max(n, 1),), -1, dtype=torch.int32, device=gpu)
    vo = torch.full((max(n, 1),), -1, dtype=torch.int32, device=gpu)
    scratch = torch.empty(lib.scn_sort_pairs_scratch_bytes(n), dtype=torch.uint8, device=gpu)
    L.check(lib.scn_sort_pairs(L.ptr(k), L.ptr(v), n, bits, L.ptr(ko), L.ptr(vo), L.ptr(scratch), L.stream()))
    torch.cuda.synchronize()
    return ko[:n].cpu().numpy().view(np.uint32), vo[:n].cpu().numpy()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1000, 4095, 4096, 4097, 5000, 12333, 51887, 150001, 600000])
@pytest.mark.parametrize("bits", [1, 6, 8, 9, 10, 18, 27, 32])
def test_sort_pairs_is_the_stable_sort(gpu, n, bits):
    rng = np.random.default_rng(n * 37 + bits)
    # few distinct keys (long equal runs: stability) mixed with full-range keys; bits above `bits` must be ignored
    lo = rng.integers(0, 1 << min(bits, 31), size=n, dtype=np.int64).astype(np.uint32)
    few = rng.integers(0, 5, size=n).astype(np.uint32) * np.uint32((1 << bits) // 7 + 1)
    keys = np.where(rng.random(n) < 0.5, lo, few).astype(np.uint32)
    junk = (rng.integers(0, 1 << 4, size=n).astype(np.uint32) << np.uint32(bits)) if bits <= 27 else np.uint32(0)
    keys_in = (keys & np.uint32((1 << bits) - 1 if bits < 32 else 0xFFFFFFFF)) | junk
    vals = rng.integers(-2**31, 2**31 - 1, size=n, dtype=np.int64).astype(np.int32)
    mask = np.uint32((1 << bits) - 1 if bits < 32 else 0xFFFFFFFF)
    order = np.argsort(keys_in & mask, kind="stable")
    for v in (vals, None):
        ko, vo = _sort(gpu, keys_in, v, bits)
        assert np.array_equal(ko, keys_in[order])
        assert np.array_equal(vo, (vals if v is not None else np.arange(n, dtype=np.int32))[order])


def test_sort_pairs_sorted_reversed_and_constant_inputs(gpu):
    for n in (4096, 70000):
        for keys in (np.arange(n, dtype=np.uint32) % (1 << 20), (np.arange(n, dtype=np.uint32)[::-1] % (1 << 20)).copy(),
                     np.full(n, 12345, np.uint32)):
            ko, vo = _sort(gpu, keys, None, 20)
            order = np.argsort(keys, kind="stable")
            assert np.array_equal(ko, keys[order]) and np.array_equal(vo, order.astype(np.int32))


def test_tile_order_is_lpt_and_stable(gpu):
    """scn_tiles_build: tile_order lists the tiles by offset count descending, equal counts in tile order (what the
    one-workgroup counting sort of round 1 produced) -- checked from the returned tile masks."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd.synthetic import make_batch
    coords, feats, size, bs, _ = make_batch(1, (128, 128, 64), 80000, dup=1.1, seed=5)      # > 4096 tiles: multi-launch form
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), bs))
    rb = x.metadata.subm_rulebook(size, 3)
    tiles = rb.tiles
    tm = tiles.tile_mask.cpu().numpy().view(np.uint32)
    order = tiles.tile_order.cpu().numpy()
    pc = np.array([bin(int(m)).count("1") for m in tm])
    assert len(tm) > 4096
    assert np.array_equal(order, np.argsort(-pc, kind="stable"))
