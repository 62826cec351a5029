"""Dataset readers and prediction-map writers (the step before / after the hot path).

Host mirror of /root/reference/isprs_dilated_random.py `load_images` :187-242, `retrieve_class_Using_RGB` /
`convert_to_class` :91-148, `retrieve_RGB_using_class` / `create_prediction_map` :118-139, 499-507, of
coffee_dilated_random.py `load_imgs_torch` / `load_images_torch` :84-125 and of contest_dilated_random.py
`convert_class` :48-69, `read_pgm` :119-143, `load_imgs_torch` :146-169.  The reference reads through
scipy.misc / gdal / skimage, none of which exist here; Pillow reads the same TIFF / JPG / PGM files.
"""
import os

import numpy as np
from PIL import Image

Image.MAX_IMAGE_PIXELS = None          # ISPRS mosaics are 6000 x 6000

# class id -> RGB of the ISPRS benchmark (isprs:118-139)
ISPRS_PALETTE = np.array([[255, 255, 255], [0, 0, 255], [0, 255, 255], [0, 255, 0], [255, 255, 0], [255, 0, 0]], dtype=np.uint8)


def img_as_float(a):
    """skimage.img_as_float for the dtypes these datasets hold."""
    a = np.asarray(a)
    if a.dtype == np.uint8:
        return a.astype(np.float64) / 255.0
    if a.dtype == np.uint16:
        return a.astype(np.float64) / 65535.0
    return a.astype(np.float64)


def imread(path):
    return np.asarray(Image.open(path))


def convert_to_class(img_label):
    """isprs:91-148, vectorised: RGB label image -> class ids 0..5 (-1 where the colour is unknown)."""
    lab = np.asarray(img_label)
    out = np.full(lab.shape[:2], -1, dtype=np.int16)
    for c, rgb in enumerate(ISPRS_PALETTE):
        out[np.all(lab[:, :, :3] == rgb, axis=2)] = c
    return out


def create_prediction_map(img_name, prob_img, size_tuple=None):
    """isprs:499-507: class map -> colour image on disk."""
    a = np.asarray(prob_img).astype(np.int64)
    Image.fromarray(ISPRS_PALETTE[a]).save(img_name)


def _potsdam_id(f):
    a, b = str(f).split("_")
    return str(f) if int(b) >= 10 else a + "_0" + b            # isprs:208-210


def load_images(path, instances, process, image_type="vaihingen"):
    """isprs:187-242: per instance an H x W x (bands + nDSM) float64 image and, for training / validate_test, its label map."""
    images, masks = [], []
    for f in instances:
        if image_type == "vaihingen":
            ndsm = img_as_float(imread(os.path.join(path, "normalized_DSM", "dsm_09cm_matching_area%s_normalized.jpg" % f)))
            rgb = img_as_float(imread(os.path.join(path, "top", "top_mosaic_09cm_area%s.tif" % f)))
            lab_file = {"validate_test": os.path.join("gts_eroded_encoding", "top_mosaic_09cm_area%s_noBoundary.tif" % f),
                        "training": os.path.join("gts_enconding", "top_mosaic_09cm_area%s.tif" % f)}
        elif image_type == "postdam":
            ndsm = img_as_float(imread(os.path.join(path, "1_DSM_normalisation", "dsm_potsdam_0%s_normalized_lastools.jpg" % _potsdam_id(f))))
            if ndsm.shape[0] != ndsm.shape[1]:                                        # isprs:211-213: one missing column
                ndsm = np.append(ndsm, np.zeros([ndsm.shape[0], 1], dtype=ndsm.dtype), axis=1)
            rgb = img_as_float(imread(os.path.join(path, "4_Ortho_RGBIR", "top_potsdam_%s_RGBIR.tif" % f)))
            lab_file = {"validate_test": os.path.join("gts_eroded_encoding", "top_potsdam_%s_label_noBoundary.tif" % f),
                        "training": os.path.join("gts_enconding", "top_potsdam_%s_label.tif" % f)}
        else:
            raise ValueError("unknown image_type " + str(image_type))
        if ndsm.ndim == 3:
            ndsm = ndsm[:, :, 0]
        lab_file["crf"] = lab_file["training"]
        images.append(np.concatenate((rgb, ndsm[:, :, None]), axis=2))
        if process in lab_file:
            lab = imread(os.path.join(path, lab_file[process]))
            masks.append(convert_to_class(lab) if lab.ndim == 3 else lab)
    return images, masks


def read_torch_ascii(path):
    """coffee:84-113 / contest:146-169: a Torch ASCII tensor dump -- 17 header lines (line 7 = 'c h w'), then the
    c*h*w values on one line; returned H x W x C float32."""
    with open(path) as fh:
        lines = fh.readlines()
    c, h, w = [int(v) for v in lines[7].split()]
    vals = np.asarray(" ".join(lines[17:]).split(), dtype=np.float32)
    return np.transpose(vals.reshape(c, h, w), (1, 2, 0))


def load_images_torch(path):
    """coffee:116-125: files sorted case-insensitively, alternating image / mask; masks rounded half up."""
    files = sorted([os.path.join(path, f) for f in os.listdir(path) if "txt" in f and f != "Thumbs.db" and "jpeg" not in f],
                   key=str.lower)
    imgs = [read_torch_ascii(f) for f in files[0::2]]
    masks = [np.floor(read_torch_ascii(f) + 0.5) for f in files[1::2]]
    return np.asarray(imgs), np.asarray(masks)


CONTEST_GRAY_TO_CLASS = {224: 6, 226: 6, 104: 0, 105: 0, 43: 5, 76: 2, 177: 3, 179: 3, 148: 1, 150: 1, 54: 4, 0: 7}   # contest:48-69


def read_pgm(path):
    """contest:119-143: ASCII PGM of grey codes -> class ids (7 = void)."""
    with open(path) as fh:
        tok = [t for line in fh if not line.startswith("#") for t in line.split()]
    assert tok[0] == "P2", "ASCII PGM expected"
    w, h = int(tok[1]), int(tok[2])
    grey = np.asarray(tok[4:4 + w * h], dtype=np.int64)
    lut = np.full(256, -1, dtype=np.int64)
    for g, c in CONTEST_GRAY_TO_CLASS.items():
        lut[g] = c
    return lut[grey].reshape(h, w)
