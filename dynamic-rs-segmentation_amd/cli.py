"""Command line: the reference's positional surface, unchanged.

isprs flavour  (isprs_dilated_random.py:1987-2042, 16 arguments):
    input_path output_path currentModelPath trainingInstances testing_instances learningRate weight_decay
    batch_size niter reference_crop_size reference_stride_crop net_type distribution_type probValues update_type process
coffee / contest flavours (coffee_dilated_random.py:1106-1150, contest_dilated_random.py:1229-1271, 14 [+ operation]):
    path_train path_test output_path currentModelPath lr wd batch niter ref_crop ref_stride net_type distribution_type
    probValues update_type [operation]

Tiles come from the ISPRS Vaihingen / Potsdam directory layouts (datasets.py: Pillow in place of the reference's
gdal / scipy.misc / skimage), from `<input_path><instance>.npz` (arrays `image` [H,W,C] float in [0,1], `label` [H,W]
uint8), or are generated when input_path is `synthetic:<H>x<W>x<C>/<dataset-name>/`.
"""
import os
import sys
import zlib

import numpy as np

from . import loops, patches as P, sampling as SP
from .net import NoComm
from .nets import resolve
from .synthetic import make_tile

ISPRS_PARAMS = ["input_path", "output_path(for model, images, etc)", "currentModelPath", "trainingInstances",
                "testing_instances", "learningRate", "weight_decay", "batch_size", "niter", "reference_crop_size",
                "reference_stride_crop",
                "net_type[dilated_icpr_original|dilated_grsl|dilated_icpr_rate6_densely|dilated_grsl_rate8|dilated8_grsl]",
                "distribution_type[single_fixed|multi_fixed|uniform|multinomial]", "probValues", "update_type [acc|loss]",
                "process [training|validate_test|generate_final_maps]"]


def print_params(list_params, argv):
    print("+" * 97)
    for i in range(1, len(argv)):
        print(list_params[i - 1] + "= " + argv[i])
    print("+" * 97)


def load_images(path, instances, process, num_classes=6, dataset=None):
    """isprs:187-242.  The ISPRS directory layouts go through datasets.load_images (Pillow instead of
    scipy.misc / gdal); `<instance>.npz` tiles and `synthetic:` paths are this build's additions."""
    from . import datasets
    if not path.startswith("synthetic:") and (os.path.isdir(os.path.join(path, "top")) or os.path.isdir(os.path.join(path, "4_Ortho_RGBIR"))):
        return datasets.load_images(path, instances, process, image_type=dataset)
    images, masks = [], []
    for f in instances:
        print(loops.BatchColors.OKBLUE + "Reading instance " + str(f) + loops.BatchColors.ENDC)
        if path.startswith("synthetic:"):
            h, w, c = [int(v) for v in path[len("synthetic:"):].split("/")[0].split("x")]
            img, lab = make_tile(h, w, c, num_classes, seed=zlib.crc32(str(f).encode()) % (2 ** 31))
        else:
            with np.load(os.path.join(path, str(f) + ".npz")) as d:
                img, lab = np.asarray(d["image"], dtype=np.float64), np.asarray(d["label"], dtype=np.uint8)
        images.append(img)
        masks.append(lab)
    return images, masks


def init_size_scores(distribution_type, values, occur_init=0):
    """isprs:2054-2064 (contest initialises patch_occur with ones, contest:1275)."""
    if distribution_type == "multi_fixed":
        n = len(values)
    elif distribution_type in ("uniform", "multinomial"):
        n = values[-1] - values[0] + 1
    else:
        return None, None, None, None
    probs = P.define_multinomial_probs(values) if distribution_type == "multinomial" else None
    return (np.zeros(n, dtype=np.float32), np.full(n, occur_init, dtype=np.int32), np.zeros(n, dtype=np.int32), probs)


def _placement(device, comm):
    """(device, comm) of this process: what the caller passed, else from the launcher's environment -- under
    `python -m torch.distributed.run --nproc-per-node N <script> ...` one rank per GPU over RCCL (dist.from_env binds the GPU and
    forms the process group before anything else touches it); a plain launch is the reference's single process on cuda:0."""
    if device is not None:
        return device, comm or NoComm()
    from .dist import from_env
    device, comm = from_env()
    return device, comm or NoComm()


def main(argv=None, device=None, comm=None):
    device, comm = _placement(device, comm)
    argv = list(sys.argv if argv is None else argv)
    if len(argv) < len(ISPRS_PARAMS) + 1:
        sys.exit("Usage: " + argv[0] + " " + " ".join(ISPRS_PARAMS))
    if comm.rank == 0:
        print_params(ISPRS_PARAMS, argv)
    (input_path, output_path, former_model_path, tr, te, lr, wd, bs, niter, ref_crop, ref_stride, net_type,
     distribution_type, prob_values, update_type, process) = argv[1:17]
    dataset = input_path[:-1].split("/")[-1].lower()
    training_instances, testing_instances = tr.split(","), te.split(",")
    lr_initial, weight_decay, batch_size, niter = float(lr), float(wd), int(bs), int(niter)
    reference_crop_size, reference_stride_crop = int(ref_crop), int(ref_stride)
    values = [int(i) for i in prob_values.split(",")]
    resolve(net_type)
    display_step = 50
    if dataset == "vaihingen":
        resample_batch = 20
    elif dataset == "postdam":
        resample_batch = 10
    else:
        print("Error! No dataset identified: ", dataset)
        resample_batch = 20
    patch_acc_loss, patch_occur, patch_chosen_values, probs = init_size_scores(distribution_type, values)

    print(loops.BatchColors.WARNING + "Reading images..." + loops.BatchColors.ENDC)
    training_data, training_labels = load_images(input_path, training_instances, process, dataset=dataset)
    testing_data, testing_labels = load_images(input_path, testing_instances, process, dataset=dataset)
    tag = os.path.join(os.getcwd(), "dataset_" + dataset + "_crop_" + str(reference_crop_size) + "_stride_" + str(reference_stride_crop))
    train_dist = test_dist = None
    if process == "training":
        train_dist = SP.create_distributions_over_classes(training_labels, reference_crop_size, reference_stride_crop)
        test_dist = SP.create_distributions_over_classes(testing_labels, reference_crop_size, reference_stride_crop)
    # the reference's cwd .npy caches (isprs:2087-2115); under data parallelism rank 0 reads / builds / writes them and
    # every rank receives rank 0's arrays (loops.rank0_cached), so no rank reads a half-written file or skips RNG draws
    rot = None
    if train_dist is not None:
        rot = loops.rank0_cached(comm, tag + "_rotation.npy", lambda: SP.create_rotation_distribution(train_dist))

    def mean_std():
        if os.path.isfile(tag + "_mean.npy"):
            return (np.load(tag + "_mean.npy"), np.load(tag + "_std.npy"))
        dist_for_stats = train_dist or SP.create_distributions_over_classes(training_labels, reference_crop_size, reference_stride_crop)
        ms = SP.dynamically_calculate_mean_and_std(training_data, dist_for_stats, crop_size=25)   # isprs:2109-2110
        np.save(tag + "_mean.npy", ms[0])
        np.save(tag + "_std.npy", ms[1])
        return ms
    mean_full, std_full = loops.rank0_call(comm, mean_std, "the mean / std caches " + tag + "_{mean,std}.npy")

    if process == "training":
        return loops.train(training_data, training_labels, train_dist, rot, testing_data, testing_labels, test_dist,
                           testing_instances, lr_initial, batch_size, niter, weight_decay, mean_full, std_full, update_type,
                           distribution_type, values, patch_acc_loss, patch_occur, patch_chosen_values, probs, resample_batch,
                           output_path, display_step, net_type, dataset, former_model_path, device=device, comm=comm)
    from .net import DilatedNet
    step = loops.step_from_model_path(former_model_path)
    sized = distribution_type in ("multi_fixed", "uniform", "multinomial")
    if sized:
        patch_acc_loss = np.load(output_path + "patch_acc_loss_step_" + str(step) + ".npy")
        patch_occur = np.load(output_path + "patch_occur_step_" + str(step) + ".npy")
    s_max = max(values)
    net = DilatedNet(net_type, training_data[0].shape[-1], 6, weight_decay, b_max=batch_size, s_max=s_max, device=device, comm=comm)
    loops.load_checkpoint(net, former_model_path)
    if process == "validate_test":
        crop = (loops.select_best_patch_size(distribution_type, values, patch_acc_loss, patch_occur, update_type, debug=True)
                if sized else int(values[0]))
        return loops.validate_test(net, testing_data, testing_labels, testing_instances, batch_size, mean_full, std_full, crop,
                                   step, output_path, comm)
    if process == "generate_final_maps":
        return loops.generate_final_maps(net, testing_data, testing_instances, batch_size, mean_full, std_full, update_type,
                                         distribution_type, values, dataset, output_path, patch_acc_loss, patch_occur, comm)
    print(loops.BatchColors.FAIL + "Process " + process + "not found!" + loops.BatchColors.ENDC)


COFFEE_PARAMS = ["path_train", "path_test", "output_path(for model, images, etc)", "currentModelPath", "learningRate",
                 "weight_decay", "batch_size", "niter", "reference_crop_size", "reference_stride_crop", "net_type",
                 "distribution_type[single_fixed|multi_fixed|uniform|multinomial]", "probValues", "update_type [acc|loss]"]
CONTEST_PARAMS = ["path", "output_path(for model, images, etc)", "currentModelPath", "learningRate", "weight_decay", "batch_size",
                  "niter", "crop_size", "stride_crop", "net_type", "distribution_type[single_fixed|multi_fixed|uniform|multinomial]",
                  "probValues", "update_type [acc|loss]", "operation [train|test]"]


def _load_stack(path, num_classes, seed0):
    """coffee `load_images_torch` (coffee:116-125) on a directory of Torch-ASCII dumps; `synthetic:<n>x<H>x<W>x<C>/` generates."""
    from . import datasets
    if path.startswith("synthetic:"):
        n, h, w, c = [int(v) for v in path[len("synthetic:"):].split("/")[0].split("x")]
        tl = [make_tile(h, w, c, num_classes, seed=seed0 + i) for i in range(n)]
        return [t[0].astype(np.float32) for t in tl], [t[1] for t in tl]
    imgs, masks = datasets.load_images_torch(path)
    return list(imgs), [np.squeeze(m).astype(np.uint8) for m in masks]


def main_coffee(argv=None, device=None, comm=None):
    """coffee_dilated_random.py:1105-1150: 2 classes, 3 bands, errorAcc_/errorOccur_/chosenValues_ side files."""
    from . import loops_indexed as LI
    device, comm = _placement(device, comm)
    argv = list(sys.argv if argv is None else argv)
    if len(argv) < len(COFFEE_PARAMS) + 1:
        sys.exit("Usage: " + argv[0] + " " + " ".join(COFFEE_PARAMS))
    if comm.rank == 0:
        print_params(COFFEE_PARAMS, argv)
    path_train, path_test, output_path, current_model, lr, wd, bs, niter, ref_crop, ref_stride, net_type, dist, pv, update_type = argv[1:15]
    values = [int(i) for i in pv.split(",")]
    resolve(net_type)
    acc, occ, chosen, probs = init_size_scores(dist, values)
    train_x, train_y = _load_stack(path_train, 2, 100)
    test_x, test_y = _load_stack(path_test, 2, 200)
    cd = LI.create_distributions_over_classes(train_y, int(ref_crop), int(ref_stride), 2)
    mean_full, std_full = LI.create_mean_and_std(train_x, int(ref_crop), int(ref_stride))
    return LI.train(train_x, train_y, test_x, test_y, cd, mean_full, std_full, output_path, current_model, float(lr), float(wd),
                    int(bs), int(niter), net_type, dist, update_type, acc, occ, chosen, probs, values, num_classes=2,
                    side_names=("errorAcc_step_", "errorOccur_step_", "chosenValues_step_"), device=device, comm=comm,
                    quantize_f16=True)                      # coffee:293: training patches pass through float16


def main_contest(argv=None, device=None, comm=None):
    """contest_dilated_random.py:1228-1313: 7 classes + void label 7, 3 bands, operation train | test."""
    from . import datasets, loops_indexed as LI
    device, comm = _placement(device, comm)
    argv = list(sys.argv if argv is None else argv)
    if len(argv) < len(CONTEST_PARAMS) + 1:
        sys.exit("Usage: " + argv[0] + " " + " ".join(CONTEST_PARAMS))
    if comm.rank == 0:
        print_params(CONTEST_PARAMS, argv)
    path, output_path, current_model, lr, wd, bs, niter, crop, stride, net_type, dist, pv, update_type, operation = argv[1:15]
    values = [int(i) for i in pv.split(",")]
    resolve(net_type)
    acc, occ, chosen, probs = init_size_scores(dist, values, occur_init=1)            # contest:1275
    if path.startswith("synthetic:"):
        h, w, c = [int(v) for v in path[len("synthetic:"):].split("/")[0].split("x")]
        (tx, ty), (ex, ey) = make_tile(h, w, c, 8, seed=11), make_tile(h, w, c, 8, seed=12)   # label 7 plays the void role
        train_x, train_y, test_x, test_y = [tx.astype(np.float32)], [ty], [ex.astype(np.float32)], [ey]
    else:
        train_x = [datasets.read_torch_ascii(path + "TelopsDatasetCityVisible_20cm_Subset.txt")]
        test_x = [datasets.read_torch_ascii(path + "TelopsDatasetCityVisible.txt")]
        train_y, test_y = [datasets.read_pgm(path + "gt8.pgm").astype(np.uint8)], [datasets.read_pgm(path + "gt_ult8.pgm").astype(np.uint8)]
    cd = LI.create_distributions_over_classes_contest(train_y[0], int(crop), int(stride), 7)       # contest:172-190
    mean_full, std_full = LI.create_mean_and_std_contest(train_x[0], cd, int(crop))                # contest:99-113
    if operation == "train":
        return LI.train(train_x, train_y, test_x, test_y, cd, mean_full, std_full, output_path, current_model, float(lr), float(wd),
                        int(bs), int(niter), net_type, dist, update_type, acc, occ, chosen, probs, values, num_classes=7,
                        void_label=7, device=device, comm=comm, flavour="contest")
    if operation == "test":
        from .net import DilatedNet
        step = loops.step_from_model_path(current_model)
        sized = dist in ("multi_fixed", "uniform", "multinomial")
        if sized:
            acc = np.load(output_path + "patch_acc_loss_step_" + str(step) + ".npy")
            occ = np.load(output_path + "patch_occur_step_" + str(step) + ".npy")
        net = DilatedNet(net_type, train_x[0].shape[-1], 7, float(wd), b_max=int(bs), s_max=max(values), device=device, comm=comm)
        loops.load_checkpoint(net, current_model)
        cs = loops.select_best_patch_size(dist, values, acc, occ, update_type, debug=True) if sized else int(values[0])
        return loops.validate_test(net, test_x, test_y, ["test"], int(bs), mean_full, std_full, cs, step, output_path, comm, ignore_label=7,
                                   flavour="contest")
    print(loops.BatchColors.FAIL + "Process " + operation + "not found!" + loops.BatchColors.ENDC)


if __name__ == "__main__":
    main()
