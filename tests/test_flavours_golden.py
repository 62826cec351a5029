"""CPU: the coffee / contest flavours of the host pipeline against goldens produced by IMPORTING the reference's own functions
(tests/golden/make_goldens.py: coffee_dilated_random.py, contest_dilated_random.py): class distributions with their quirks,
mean / std, flip-by-index sampling (shift-back at the borders), window enumeration, Torch-ASCII and PGM readers."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.fixture(scope="module")
def coffee():
    return np.load(os.path.join(HERE, "golden", "coffee.npz"))


@pytest.fixture(scope="module")
def contest():
    return np.load(os.path.join(HERE, "golden", "contest.npz"))


def test_coffee_class_distribution_and_statistics(coffee):
    from drs_amd import loops_indexed as LI
    dist = LI.create_distributions_over_classes([coffee["mask"][0], coffee["mask"][1]], 9, 4, 2)
    np.testing.assert_array_equal(np.asarray(dist), coffee["dist"])
    mean, std = LI.create_mean_and_std(coffee["data"], 9, 4)
    np.testing.assert_allclose(mean, coffee["mean"], rtol=2e-6)
    np.testing.assert_allclose(std, coffee["std"], rtol=2e-6)


def test_contest_class_distribution_quirks_and_statistics(contest):
    from drs_amd import loops_indexed as LI
    dist = LI.create_distributions_over_classes_contest(contest["lab"], 9, 4)
    np.testing.assert_array_equal(np.asarray(dist)[:, 1:], contest["dist"])
    lab = contest["lab"]
    # the quirks are really in the golden: uniform non-void windows are dropped, and the vote ignores the highest class present
    assert not any((lab[x:x + 9, y:y + 9] == 3).all() for x, y in contest["dist"])
    assert ((lab[0:9, 0:9] == 3).all())
    mean, std = LI.create_mean_and_std_contest(contest["data"], dist, 9)
    np.testing.assert_allclose(mean, contest["mean"], rtol=2e-6)
    np.testing.assert_allclose(std, contest["std"], rtol=2e-6)


@pytest.mark.parametrize("S", [9, 13])
def test_coffee_flip_by_index_sampler_host_reference(coffee, S):
    """oracle/host_ref.indexed_create_patches restates coffee:241-293 (float16 cast included) and contest:192-254."""
    from oracle import host_ref as H
    dist = [(int(k), int(i), int(j)) for k, i, j in coffee["dist"]]
    p, c, _ = H.indexed_create_patches(coffee["data"], coffee["mask"][..., 0], S, dist, coffee["shuffle"], float16=True)
    assert p.dtype == np.float16
    np.testing.assert_array_equal(p, coffee["patches16_%d" % S])
    np.testing.assert_array_equal(c, coffee["classes_%d" % S][..., 0])       # the reference keeps the mask's trailing unit axis
    pn = H.normalize_images_f16(p, coffee["mean"], coffee["std"])
    np.testing.assert_array_equal(pn, coffee["normalized16_%d" % S])


@pytest.mark.parametrize("S", [9, 14])
def test_contest_sampler_with_void_mask_host_reference(contest, S):
    from oracle import host_ref as H
    dist = [(0, int(i), int(j)) for i, j in contest["dist"]]
    p, c, m = H.indexed_create_patches(contest["data"][None], contest["lab"][None], S, dist, contest["shuffle"], void_label=7)
    np.testing.assert_array_equal(p, contest["patches_%d" % S])
    np.testing.assert_array_equal(c, contest["classes_%d" % S])
    np.testing.assert_array_equal(m, contest["masks_%d" % S])


def test_window_enumeration_of_both_flavours(coffee, contest):
    from drs_amd import patches as P
    for g, (h, w), flavour in ((coffee, (40, 40), "coffee"), (contest, (60, 50), "contest")):
        for tag in "abc":
            s, st, idx, bs = [int(v) for v in g["win_args_" + tag]]
            np.testing.assert_array_equal(P.window_positions(h, w, s, st, idx, bs, flavour), g["win_pos_" + tag].astype(np.int64))
    # contest:275 divides by the row count: on this 60 x 50 tile batch 1 does not start where batch 0 ended (the isprs form does)
    s, st, idx, bs = [int(v) for v in contest["win_args_c"]]
    assert P.window_start(60, 50, s, st, idx, bs, "contest") != P.window_start(60, 50, s, st, idx, bs)
    # contest's void mask per window (contest:300-305)
    s, st, idx, bs = [int(v) for v in contest["win_args_b"]]
    pos = P.window_positions(60, 50, s, st, idx, bs, "contest")
    want = np.stack([contest["lab"][x:x + s, y:y + s] != 7 for x, y in pos])
    np.testing.assert_array_equal(want, contest["win_mask_b"])


def test_torch_ascii_reader_against_the_reference(coffee, tmp_path):
    from drs_amd import datasets
    from torch_ascii_fixture import write_torch_ascii
    for nm, seed in zip(coffee["torch_names"], coffee["torch_seeds"]):
        nm = str(nm)
        write_torch_ascii(str(tmp_path / nm), seed=int(seed), c=1 if "mask" in nm else 3, h=500, w=500, is_mask="mask" in nm)
    imgs, masks = datasets.load_images_torch(str(tmp_path) + "/")
    assert tuple(imgs.shape) == tuple(coffee["torch_img_shape"]) and tuple(masks.shape) == tuple(coffee["torch_mask_shape"])
    np.testing.assert_array_equal(imgs[:, ::37, ::41, :], coffee["torch_img_sample"])
    np.testing.assert_array_equal(masks[:, ::37, ::41, :], coffee["torch_mask_sample"])
    np.testing.assert_allclose(imgs.astype(np.float64).sum(axis=(1, 2)), coffee["torch_img_sum"], rtol=1e-12)
    np.testing.assert_array_equal(masks.astype(np.float64).sum(axis=(1, 2, 3)), coffee["torch_mask_sum"])


def test_pgm_reader_against_the_reference(contest):
    from drs_amd import datasets
    np.testing.assert_array_equal(datasets.read_pgm(os.path.join(HERE, "golden", "contest_gt.pgm")), contest["pgm"])
    assert float(contest["acc_quirk"][0]) == 0.0 and int(contest["acc_quirk_track_sum"][0]) == 0     # contest:339 never counts
