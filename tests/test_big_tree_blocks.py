"""MA trees beyond one ballot (more than 64 decision nodes or leaves after pruning) are evaluated by the wave loop from a BLOCK form
(jxl_coder_amd/csrc/dev_modular.h: big_tree_build — at most 63 decision nodes and 64 exits per block, cut breadth first).  The builder is
device code that also compiles for the CPU: tests/emul runs it on seeded random trees (static channel / stream decisions, chains, leaves
with multipliers and offsets) and compares the block-form walk (big_tree_eval) with the plain tree walk on random property vectors.  What
the MI355X does with the blocks — lane i decides node i, lane j tests exit j — is covered by the -m gpu parity test on a reference-made RGBA
photograph (tests/test_gpu_parity.py: test_alpha_streams_with_a_tree_of_hundreds_of_leaves_run_from_its_block_form)."""
import ctypes as C

import pytest


@pytest.fixture(scope="module")
def lib(emul):
    import os
    from conftest import ROOT
    L = C.CDLL(os.path.join(ROOT, "tests", "emul", "libjxlemul.so"))
    L.emul_bigtree_selftest.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32 * 3)]
    return L


@pytest.mark.parametrize("nodes", [1, 10, 63, 64, 65, 127, 200, 458, 1000])
def test_block_form_walk_equals_tree_walk(lib, nodes):
    st = (C.c_int32 * 3)()
    built = 0
    for seed in range(40):
        r = lib.emul_bigtree_selftest(seed, nodes, 30208, 400, seed & 1, C.byref(st))
        assert r in (0, -1, -3), (seed, nodes, r)          # 0: every walk agrees; -1: does not fit the area; -3: deeper than the builder's stack (the decoder takes the serial walker)
        if r == 0:
            built += 1
            assert st[1] <= nodes and st[2] == st[1] + st[0]      # exits = decision nodes + 1 per block ... summed: nodes + blocks
            if st[1] > 63:
                assert st[0] > 1
    assert built >= 30


def test_block_form_reports_an_area_that_is_too_small(lib):
    st = (C.c_int32 * 3)()
    assert lib.emul_bigtree_selftest(5, 458, 2000, 10, 0, C.byref(st)) == -1
    assert lib.emul_bigtree_selftest(5, 458, 30208, 400, 0, C.byref(st)) == 0
