"""CPU: the product's preprocessing helpers and size scoring against the reference-generated goldens."""
import os
import random

import numpy as np

from drs_amd import sampling as SP
from drs_amd import loops


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_class_distribution_rotation_superbatch_meanstd(golden_dir):
    g = _g(golden_dir, "sampling.npz")
    labels = [g["lab0"], g["lab1"]]
    dist = SP.create_distributions_over_classes(labels, 25, 5)
    for k in range(6):
        np.testing.assert_array_equal(np.asarray(dist[k], dtype=np.int64).reshape(-1, 3), g["class%d" % k])
    np.random.seed(99)
    rot = SP.create_rotation_distribution(dist)
    for k in range(6):
        np.testing.assert_array_equal(rot[k], g["rot%d" % k])
    random.seed(21)
    np.random.seed(21)
    sb = SP.select_super_batch_instances(dist, rot, batch_size=7, super_batch=5)
    np.testing.assert_array_equal(sb, g["super_batch"])
    mean, std = SP.dynamically_calculate_mean_and_std([g["data0"], g["data1"]], dist, 25)
    np.testing.assert_array_equal(mean, g["mean_full"])
    np.testing.assert_array_equal(std, g["std_full"])


def test_best_patch_size(golden_dir):
    g = _g(golden_dir, "best_size.npz")
    for i in range(4):
        for mode in ("loss", "acc"):
            sums, cnt = g["case%d_sums" % i].copy(), g["case%d_cnt" % i].copy()
            ch = np.zeros(len(sums), dtype=np.int32)
            best = loops.select_best_patch_size(str(g["case%d_dist" % i]), list(g["case%d_vals" % i]), sums, cnt, mode, ch)
            assert best == int(g["case%d_%s" % (i, mode)][0])
            np.testing.assert_array_equal(cnt, g["case%d_%s_occur_after" % (i, mode)])
            np.testing.assert_array_equal(ch, g["case%d_%s_chosen" % (i, mode)])


def test_plan_tables_and_aliases():
    from drs_amd.nets import Plan, resolve, known_net_types
    assert resolve("dilated8_grsl") == resolve("dilated_grsl_rate8") == "dilated_grsl_rate8"
    assert {"dilated_icpr_original", "dilated_grsl", "dilated_icpr_rate6_densely", "dilated_grsl_rate8"} <= set(known_net_types())
    p = Plan("dilated_grsl_rate8", 5, 6)
    assert p.n_params == 2091590 and p.mac_per_pixel() == 2090304
    assert [(L.pad_b, L.pad_a) for L in p.layers][2] == (4, 5)             # 4x4 kernel at rate 3: asymmetric SAME
    assert Plan("dilated_grsl", 5, 6).n_params == 1390790
    assert Plan("dilated_icpr_original", 3, 6).n_params == 1387590
    d = Plan("dilated_icpr_rate6_densely", 4, 2)
    assert d.n_params == 816578 and d.concat_off == [0, 32, 64, 128, 192, 320] and d.concat_halo == 6
    try:
        resolve("no_such_net")
        assert False
    except ValueError:
        pass
