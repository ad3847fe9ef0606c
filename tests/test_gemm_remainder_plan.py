"""CPU: the remainder-round plans of the ring GEMMs (csrc/gemm.hip: ring_split, SkPlan) as the launcher and the kernels compute them
(`mantis_gemm_remainder_plan`, host arithmetic only -- the library answers for a 256-CU device when there is no GPU).  Invariants that make
the K-split correct BY CONSTRUCTION, whatever order the hardware runs the workgroups in: every K-step of every remainder tile is covered by
exactly one workgroup, no workgroup spans tiles, slabs are unique and fit the workspace; plus the placement the balanced round's speed rests
on (a tail sits on the XCD of its range's head, shortest head first per XCD)."""
import ctypes

import numpy as np
import pytest


def plan(M, N, K, cus=0):
    from mantis_amd import _lib
    L = _lib.load()
    buf = np.zeros(8 + 4 * 1024, dtype=np.int32)
    n = L.mantis_gemm_remainder_plan(M, N, K, cus, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    assert 8 <= n <= buf.size
    head = dict(zip(("tiles", "full", "rem", "S", "units", "tails", "nwg", "nk"), buf[:8].tolist()))
    return head, buf[8:n].reshape(-1, 4), L


# the step's shapes with an incomplete last round on 256 CUs (+ budgets, ragged and long-K cases)
SHAPES = [(5624, 4096, 4096, 0), (5624, 4096, 6144, 0), (5624, 4096, 14336, 0), (5624, 4096, 28672, 0), (5624, 6144, 4096, 0),
          (6144, 4096, 5624, 0), (4096, 14336, 5624, 0), (2000, 2100, 2048, 0), (5624, 3840, 4096, 0), (5624, 4096, 28672, 240),
          (5624, 4096, 28672, 200), (5624, 4096, 14336, 97), (4608, 1152, 4304, 0), (8192, 4096, 16384, 0), (5000, 4096, 100000, 0)]


@pytest.mark.parametrize("M,N,K,cus", SHAPES)
def test_every_k_step_of_every_remainder_tile_is_covered_once(M, N, K, cus):
    h, recs, L = plan(M, N, K, cus)
    budget = L.mantis_gemm_cu_budget(cus)
    assert h["tiles"] == -(-M // 256) * -(-N // 256) and h["nk"] == -(-K // 64)
    if h["S"] == 1:
        assert h["nwg"] == 0 and h["full"] == h["tiles"]
        return
    assert h["full"] % budget == 0 and h["rem"] == h["tiles"] - h["full"] and 0 < h["rem"] < budget
    cover = np.zeros((h["rem"], h["nk"]), dtype=np.int32)
    slabs = set()
    live = [r for r in recs if r[0] >= 0]
    for tl, t0, t1, slab in live:
        assert 0 <= tl < h["rem"] and 0 <= t0 < t1 <= h["nk"], (tl, t0, t1)      # inside ONE tile, never empty
        cover[tl, t0:t1] += 1
        assert slab not in slabs
        slabs.add(int(slab))
    assert (cover == 1).all(), "a K-step of a remainder tile is covered twice or not at all"
    ws_slabs = (L.mantis_gemm_workspace_bytes(0, 0, 0) - 4096) // (256 * 256 * 4)
    assert max(slabs) < ws_slabs, "slab index outside the caller's workspace"
    if h["units"] == 0:                                        # equal split: S parts per tile, one round of workgroups
        assert len(recs) == h["S"] * h["rem"] <= budget
    else:                                                      # balanced round
        U = h["units"]
        assert U == budget and h["tails"] % 8 == 0 and len(recs) == U + h["tails"]
        heads, tails = recs[:U], recs[U:]
        assert (heads[:, 0] >= 0).all()
        lens = heads[:, 2] - heads[:, 1]
        # a range = head (+ tail): every CU gets the same number of K-steps (+- the snapping of boundaries next to a tile boundary)
        total = lens.astype(np.int64).copy()
        head_of_tile_end = {}
        for c, (tl, t0, t1, slab) in enumerate(heads):
            assert slab == c
            if t1 == h["nk"]:
                head_of_tile_end[int(tl)] = c
        for i, (tl, t0, t1, slab) in enumerate(tails):
            if tl < 0:
                continue
            assert t0 == 0 and slab == U + tl - 1
            c = head_of_tile_end[int(tl) - 1]                  # the range that ended the previous tile continues here
            total[c] += t1 - t0
            assert i % 8 == c % 8, "a tail must be dealt to the XCD (workgroup index mod 8) its head ran on"
        w = h["rem"] * h["nk"] / U
        assert total.min() >= w - 9 and total.max() <= w + 9, (total.min(), total.max(), w)
        # per XCD: shortest head first
        for x in range(8):
            hl = [int(lens[head_of_tile_end[int(tl) - 1]]) for i, (tl, _, _, _) in enumerate(tails) if tl >= 0 and i % 8 == x]
            assert hl == sorted(hl)
        # chosen only where it pays: it saves at least 24 K-steps per CU over the equal split
        assert h["nk"] // h["S"] - int(w) >= 24


def test_balanced_round_is_chosen_for_the_long_k_shapes_of_the_step_only():
    for (M, N, K), balanced in {(5624, 4096, 28672): True, (5624, 4096, 14336): True, (5624, 4096, 4096): False, (5624, 4096, 6144): False,
                                (5624, 6144, 4096): False, (6144, 4096, 5624): False}.items():
        h, _, _ = plan(M, N, K)
        assert (h["units"] != 0) == balanced, (M, N, K, h)
    h, recs, _ = plan(5624, 4096, 28672)
    assert h["rem"] == 96 and h["units"] == 256 and h["S"] == 2 and h["tails"] == 256      # 64 tails on two XCDs, 8 x 32 slots
    assert int((recs[256:, 0] >= 0).sum()) == 64
