"""The coffee / contest flavours of the training loop.

Host mirror of /root/reference/coffee_dilated_random.py `main` :1105-1369 (there the loop lives in main) and
contest_dilated_random.py `train` :955-1150: no super-batch, no rotation / noise; the permutation runs over 3*N indices and
the index range encodes the flip ([0,N) as is, [N,2N) left-right, [2N,3N) up-down: coffee:283-291, contest:241-252); LR decay
factor 0.1; the 'loss' score is the plain batch loss; contest masks the void label out of the loss and the accuracy
(contest:235-239, 881-901) and starts `patch_occur` at ones.  coffee's training patches pass through float16 (coffee:293) and
are normalised in that array: `quantize_f16` (drs_crop_normalize), pinned by tests/golden/coffee.npz.
"""
import random

import numpy as np
import torch

from . import loops, metrics as MT, patches as P
from .dist import shard_slice
from .net import DilatedNet, NoComm


def create_distributions_over_classes(labels, crop_size, stride_crop, num_classes):
    """coffee:358-372 / contest:~300: windows on a stride grid that fit entirely (NO shift-back), bucketed by majority
    class, buckets concatenated in class order.  Rows (map, x, y)."""
    buckets = [[] for _ in range(num_classes + 1)]
    for k, lab in enumerate(labels):
        lab = np.squeeze(np.asarray(lab))
        h, w = lab.shape
        for i in range(0, h, stride_crop):
            for j in range(0, w, stride_crop):
                win = lab[i:i + crop_size, j:j + crop_size]
                if win.shape == (crop_size, crop_size):
                    c = int(np.argmax(np.bincount(win.astype(int).flatten())))
                    buckets[min(c, num_classes)].append((k, i, j))
    return [w for b in buckets[:num_classes] for w in b]


def create_mean_and_std(training_data, crop_size, stride_crop):
    """coffee:352-355 -> create_crops_stride :176-238 -> compute_image_mean :77-81: every fitting window on the stride
    grid (odd crop sizes advance by stride+1 on every second step) plus its two flips; mean over everything, standard
    deviation (ddof=1) across crops AT PIXEL (0,0).  The flips leave the mean unchanged and put two other corners of the
    window at (0,0), so nothing is materialised."""
    means, corners = [], []
    for img in training_data:
        h, w = img.shape[0], img.shape[1]
        j, cx = 0, 0
        while j < h:
            k, cy = 0, 0
            while k < w:
                if j + crop_size <= h and k + crop_size <= w:
                    win = img[j:j + crop_size, k:k + crop_size, :]
                    means.append(win.mean(axis=(0, 1)))
                    corners += [win[0, 0], win[0, -1], win[-1, 0]]
                k += stride_crop + 1 if (crop_size % 2 != 0 and cy % 2 != 0) else stride_crop
                cy += 1
            j += stride_crop + 1 if (crop_size % 2 != 0 and cx % 2 != 0) else stride_crop
            cx += 1
    return np.mean(means, axis=0), np.std(np.asarray(corners), axis=0, ddof=1)


def create_distributions_over_classes_contest(labels, crop_size, stride_crop, num_classes=7):
    """contest:172-190, for its single label map, quirks included: a window is dropped when the HIGHEST class id present in it
    fills it (`count[-1] == crop^2`: meant for all-void windows, true for any uniform window), and the majority vote runs over
    `count[:-1]`, i.e. without that highest class.  Rows (0, x, y), buckets concatenated in class order."""
    buckets = [[] for _ in range(num_classes)]
    lab = np.squeeze(np.asarray(labels))
    h, w = lab.shape
    for i in range(0, h, stride_crop):
        for j in range(0, w, stride_crop):
            win = lab[i:i + crop_size, j:j + crop_size]
            if win.shape == (crop_size, crop_size):
                count = np.bincount(win.astype(int).flatten())
                if count[-1] == crop_size * crop_size:
                    continue
                buckets[int(np.argmax(count[:-1]))].append((0, i, j))
    return [w_ for b in buckets for w_ in b]


def create_mean_and_std_contest(data, class_distribution, crop_size):
    """contest:99-113: mean over the class-distribution windows, standard deviation (ddof=1) across them AT PIXEL (0,0)."""
    means = [data[x:x + crop_size, y:y + crop_size, :].mean(axis=(0, 1), dtype=np.float64) for (_, x, y) in class_distribution]
    corners = np.asarray([data[x, y, :] for (_, x, y) in class_distribution])
    return np.mean(means, axis=0), np.std(corners, axis=0, ddof=1)


def train(training_data, training_labels, test_data, test_labels, class_distribution, mean_full, std_full, output_path,
          current_model, lr_initial, weight_decay, batch_size, niter, net_type, distribution_type, update_type, patch_acc_loss,
          patch_occur, patch_chosen_values, probs, values, *, num_classes, void_label=-1, side_names=None, device="cuda:0",
          comm=None, display_step=50, quiet_sizes=False, quantize_f16=False, flavour="isprs"):
    comm = comm or NoComm()
    if batch_size % comm.world:
        raise ValueError("batch_size must be divisible by the number of ranks")
    loops.sync_rng(comm)                 # every rank walks the same permutation and draws the same sizes
    say = (lambda *a: print(*a)) if comm.rank == 0 else (lambda *a: None)
    side = side_names or ("patch_acc_loss_step_", "patch_occur_step_", "patch_chosen_values_step_")
    channels = training_data[0].shape[-1]
    N = len(class_distribution)
    total = 3 * N
    dist_arr = np.asarray(class_distribution, dtype=np.int64)
    b_local = batch_size // comm.world
    sl = shard_slice(batch_size, comm.rank, comm.world)
    sized = distribution_type in ("multi_fixed", "uniform", "multinomial")
    s_max = int(max(values)) if sized else int(values[0])
    net = DilatedNet(net_type, channels, num_classes, weight_decay, b_max=b_local, s_max=s_max, device=device, comm=comm,
                     lr_decay_factor=0.1)                                            # coffee:1228, contest:1021
    train_pool = P.TilePool(training_data, [np.squeeze(l) for l in training_labels], device)
    test_pool = P.TilePool(test_data, [np.squeeze(l) for l in test_labels], device)
    loops.check_training_labels(train_pool, num_classes, void_label if void_label >= 0 else None)
    shuffle = np.asarray(random.sample(range(total), total))
    current_iter = 1
    if current_model is not None and "model" in current_model:
        current_iter = loops.step_from_model_path(current_model)
        if sized:
            patch_acc_loss = np.load(output_path + side[0] + str(current_iter) + ".npy")
            patch_occur = np.load(output_path + side[1] + str(current_iter) + ".npy")
            patch_chosen_values = np.load(output_path + side[2] + str(current_iter) + ".npy")
        loops.load_checkpoint(net, current_model)
    else:
        say("Model totally initialized!")

    def save(step):
        if comm.rank == 0:
            np.savez(output_path + "model-" + str(step) + ".npz", **net.state_dict())
            if sized:
                np.save(output_path + side[0] + str(step) + ".npy", patch_acc_loss)
                np.save(output_path + side[1] + str(step) + ".npy", patch_occur)
                np.save(output_path + side[2] + str(step) + ".npy", patch_chosen_values)

    def test(step):
        cur = (loops.select_best_patch_size(distribution_type, values, patch_acc_loss, patch_occur, update_type, patch_chosen_values,
                                            debug=comm.rank == 0) if sized else int(values[0]))
        return loops.validate_test(net, test_data, test_labels, list(range(len(test_data))), batch_size, mean_full, std_full, cur,
                                   step, output_path, comm, pool=test_pool, ignore_label=void_label, flavour=flavour)

    it, epoch_mean = 0, 0.0
    epoch_cm = np.zeros((num_classes, num_classes), dtype=np.uint32)
    step = current_iter
    for step in range(current_iter, niter + 1):
        cur_size, cur_size_int = P.draw_patch_size(distribution_type, values, probs)
        if not quiet_sizes:
            say(cur_size)
        shuffle, batch, it = P.select_batch(shuffle, batch_size, it, total)
        if step - current_iter < 3:
            comm.agree((cur_size, batch[0], batch[-1], it), "patch size / batch indices at step %d" % step)
        flip = np.where(batch >= 2 * N, 1, np.where(batch >= N, 2, 0))              # kernel codes: 1 = flipud, 2 = fliplr
        rows = dist_arr[batch % N]
        aug = P.Augmentation(b_local)
        aug.flip = flip[sl].astype(np.int32)
        P.crop_to_net(net, train_pool, rows[sl], cur_size, mean_full, std_full, aug, void_label=void_label, quantize_f16=quantize_f16)
        M = b_local * cur_size * cur_size
        if void_label >= 0:                      # masked mean: the loss averages over the unmasked pixels of the global batch
            cnt = net.acc_mask[:M].sum(dtype=torch.float64).reshape(1)
            comm.all_reduce_sum(cnt)
            npix = max(1.0, float(cnt.item()))
            net.loss_mask[:M].copy_(net.acc_mask[:M])
            out = net.train_step(b_local, cur_size, lr_initial, use_loss_mask=True, use_acc_mask=True, global_pixels=npix)
        else:
            out = net.train_step(b_local, cur_size, lr_initial, use_acc_mask=False)
        cm = out["conf"].cpu().numpy().astype(np.uint32)
        loss = net.loss_value(out["loss_parts"])
        acc, oa, acc_norm = MT.overall_and_normalized(cm)
        epoch_mean += acc
        epoch_cm += cm
        if sized:
            patch_acc_loss[cur_size_int] += loss if update_type == "loss" else acc_norm
            patch_occur[cur_size_int] += 1
        if step != 0 and step % display_step == 0:
            say("Iter " + str(step) + " -- Training Minibatch: Loss= " + "{:.6f}".format(loss) +
                " Absolut Right Pred= " + str(int(acc)) + " Overall Accuracy= " + "{:.4f}".format(oa) +
                " Normalized Accuracy= " + "{:.4f}".format(acc_norm) + " Confusion Matrix= " + loops._cm_str(cm))
        if step % loops.EPOCH_NUMBER == 0:
            _, _, na = MT.overall_and_normalized(epoch_cm)
            say("-- Iter " + str(step) + " -- Training Epoch:" +
                " Overall Accuracy= " + "{:.6f}".format(epoch_mean / max(1.0, float(np.sum(epoch_cm)))) +
                " Normalized Accuracy= " + "{:.6f}".format(na) + " Confusion Matrix= " + loops._cm_str(epoch_cm))
            epoch_mean, epoch_cm = 0.0, np.zeros((num_classes, num_classes), dtype=np.uint32)
        if step != 0 and step % loops.VAL_INTERVAL == 0:
            save(step)
            test(step)
    say("Optimization Finished!")
    save(step)
    test(step)
    return net
