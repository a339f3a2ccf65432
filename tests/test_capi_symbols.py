"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares.
No compute call is made here (there is no GPU); the product must FAIL LOUDLY, never fall back."""
import ctypes as C

import pytest

import sr_livo_amd as srl
from sr_livo_amd import capi


def test_library_exports_every_declared_symbol():
    declared = srl.declared_symbols()
    assert len(declared) >= 45
    lib = srl.load_library()
    missing = [n for n in declared if not hasattr(lib, n)]
    assert missing == []


def test_struct_layouts_match_the_header():
    # sizes the C compiler produced for the structs of include/srlivo_hip.h (checked in csrc at build time too)
    assert C.sizeof(capi.NormalEq) == 36 * 8 + 6 * 8 + 8 + 4 + 4 + 8 + 8 + 4 + 4
    assert C.sizeof(capi.Frame) == (4 + 3 + 3 + 9 + 3) * 8 + 8
    assert C.sizeof(capi.Timing) == 4 * 4 + 8 + 3 * 8 + 2 * 8 + 3 * 8 + 8           # + sum_passes (round 3)


def test_default_options_are_the_effective_yaml_values():
    o = srl.default_opts()
    assert (o.size_voxel_map, o.num_iters_icp, o.min_number_neighbors, o.voxel_neighborhood) == (1.0, 5, 20, 1)
    assert (o.max_number_neighbors, o.max_dist_to_plane_icp, o.max_num_residuals) == (20, 0.3, 600)
    assert (o.threshold_orientation_norm, o.threshold_translation_norm) == (0.1, 0.01)
    assert (o.weight_alpha, o.weight_neighborhood, o.power_planarity) == (0.9, 0.1, 2.0)


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(srl.SrlError) as ei:
        srl.Context(0)
    assert ei.value.status == capi.SRL_ERR_NO_DEVICE
    with pytest.raises(srl.SrlError):
        srl.Lio(0)


def test_shard_range_partitions_contiguously():
    for n in (0, 1, 7, 4096, 65536, 262144):
        for R in (1, 2, 3, 4, 8):
            pos = 0
            for r in range(R):
                b, c = srl.shard_range(n, R, r)
                assert b == pos and c >= 0
                pos += c
            assert pos == n


def test_shard_budget_reproduces_the_sequential_cutoff():
    # optimize.cpp:107: the loop stops once num_residuals >= max_num_residuals
    assert srl.shard_budget(600, [100, 600, 5], 0) == (600, 0)
    assert srl.shard_budget(600, [100, 600, 5], 1) == (500, 0)
    b, m = srl.shard_budget(600, [100, 600, 5], 2)
    assert m == 2 and b <= 0
    # class default -1: only the first keypoint of the sweep is visited
    assert srl.shard_budget(-1, [3, 3], 0)[1] == 1
    assert srl.shard_budget(-1, [3, 3], 1)[1] == 2
