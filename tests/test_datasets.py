"""CPU: dataset readers on files synthesised in the reference datasets' formats and directory layouts."""
import os

import numpy as np
from PIL import Image

from drs_amd import datasets as D


def test_vaihingen_layout_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    root = str(tmp_path)
    for d in ("normalized_DSM", "top", "gts_enconding", "gts_eroded_encoding"):
        os.makedirs(os.path.join(root, d))
    rgb = rng.integers(0, 256, size=(40, 50, 3), dtype=np.uint8)
    dsm = rng.integers(0, 256, size=(40, 50), dtype=np.uint8)
    cls = rng.integers(0, 6, size=(40, 50))
    Image.fromarray(rgb).save(os.path.join(root, "top", "top_mosaic_09cm_area7.tif"))
    Image.fromarray(dsm).save(os.path.join(root, "normalized_DSM", "dsm_09cm_matching_area7_normalized.jpg"), quality=100)
    Image.fromarray(D.ISPRS_PALETTE[cls]).save(os.path.join(root, "gts_enconding", "top_mosaic_09cm_area7.tif"))
    Image.fromarray(cls.astype(np.uint8)).save(os.path.join(root, "gts_eroded_encoding", "top_mosaic_09cm_area7_noBoundary.tif"))
    imgs, masks = D.load_images(root, ["7"], "training", "vaihingen")
    assert imgs[0].shape == (40, 50, 4) and imgs[0].dtype == np.float64
    np.testing.assert_array_equal(imgs[0][:, :, :3], rgb / 255.0)
    assert np.abs(imgs[0][:, :, 3] - dsm / 255.0).max() < 0.05          # JPEG
    np.testing.assert_array_equal(masks[0], cls)                         # RGB-coded labels -> class ids
    imgs, masks = D.load_images(root, ["7"], "validate_test", "vaihingen")
    np.testing.assert_array_equal(masks[0], cls)
    imgs, masks = D.load_images(root, ["7"], "generate_final_maps", "vaihingen")
    assert masks == []
    out = os.path.join(root, "pred.png")
    D.create_prediction_map(out, cls)
    np.testing.assert_array_equal(D.convert_to_class(np.asarray(Image.open(out))), cls)


def test_potsdam_names_and_dsm_padding(tmp_path):
    rng = np.random.default_rng(1)
    root = str(tmp_path)
    for d in ("1_DSM_normalisation", "4_Ortho_RGBIR", "gts_enconding"):
        os.makedirs(os.path.join(root, d))
    rgbir = rng.integers(0, 256, size=(32, 32, 4), dtype=np.uint8)
    Image.fromarray(rgbir, mode="RGBA").save(os.path.join(root, "4_Ortho_RGBIR", "top_potsdam_2_7_RGBIR.tif"))
    Image.fromarray(rng.integers(0, 256, size=(32, 31), dtype=np.uint8)).save(
        os.path.join(root, "1_DSM_normalisation", "dsm_potsdam_02_07_normalized_lastools.jpg"))
    cls = rng.integers(0, 6, size=(32, 32))
    Image.fromarray(D.ISPRS_PALETTE[cls]).save(os.path.join(root, "gts_enconding", "top_potsdam_2_7_label.tif"))
    imgs, masks = D.load_images(root, ["2_7"], "training", "postdam")
    assert imgs[0].shape == (32, 32, 5)                                  # RGBIR + nDSM padded by one column (isprs:211-213)
    np.testing.assert_array_equal(imgs[0][:, :, :4], rgbir / 255.0)
    assert np.all(imgs[0][:, 31, 4] == 0)
    np.testing.assert_array_equal(masks[0], cls)
    assert D._potsdam_id("6_12") == "6_12" and D._potsdam_id("6_7") == "6_07"


def test_torch_ascii_and_pgm(tmp_path):
    rng = np.random.default_rng(2)
    t = rng.uniform(size=(3, 5, 6)).astype(np.float32)
    p = os.path.join(str(tmp_path), "a_img.txt")
    with open(p, "w") as fh:
        fh.write("".join("h%d\n" % i for i in range(7)) + "3 5 6\n" + "".join("h%d\n" % i for i in range(8, 17)))
        fh.write(" ".join(repr(float(v)) for v in t.reshape(-1)) + "\n")
    a = D.read_torch_ascii(p)
    assert a.shape == (5, 6, 3)
    np.testing.assert_allclose(a, np.transpose(t, (1, 2, 0)), rtol=1e-6)
    m = (rng.uniform(size=(1, 5, 6)) > 0.5).astype(np.float32) * 0.9
    with open(os.path.join(str(tmp_path), "a_mask.txt"), "w") as fh:
        fh.write("".join("h%d\n" % i for i in range(7)) + "1 5 6\n" + "".join("h%d\n" % i for i in range(8, 17)))
        fh.write(" ".join(repr(float(v)) for v in m.reshape(-1)) + "\n")
    imgs, masks = D.load_images_torch(str(tmp_path) + "/")
    assert imgs.shape == (1, 5, 6, 3) and masks.shape == (1, 5, 6, 1) and set(np.unique(masks)) <= {0.0, 1.0}
    grey = np.array([[224, 104, 43], [76, 177, 0]])
    pg = os.path.join(str(tmp_path), "l.pgm")
    with open(pg, "w") as fh:
        fh.write("P2\n# comment\n3 2\n255\n" + "\n".join(" ".join(str(v) for v in r) for r in grey) + "\n")
    np.testing.assert_array_equal(D.read_pgm(pg), [[6, 0, 5], [2, 3, 7]])
