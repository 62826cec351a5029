"""Step loops: training, patch validation, whole-tile sliding-window inference.

Host mirror of /root/reference/isprs_dilated_random.py `train` :1621-1851, `validation` :1569-1618,
`validate_test` :1241-1344 and `generate_final_maps` :1854-1957 (control flow, constants, log line formats and
side files), with every per-pixel operation on the device.  Differences that are deliberate:

  * the reference reads `loss`, `pred_up` back every step and runs a per-pixel Python loop on them
    (calc_accuracy_by_crop, isprs:1754); here the confusion matrix and the loss stay on the device and are read back
    ONE STEP LATE (while the next step runs), so the host never stalls the GPU.  The size-score update is a sum, so
    the lag does not change any result; it is flushed before every display / save / validation point;
  * the TensorFlow checkpoint becomes `model-<step>.npz` (same names, same `-<step>` resume convention, isprs:1708-1715);
  * under data parallelism every rank runs the same host code with the same RNG streams and takes its slice of the batch.
"""
import datetime
import math
import os
import random

import numpy as np
import torch

from . import metrics as MT
from . import patches as P
from . import sampling as SP
from .dist import shard_slice
from .net import DilatedNet, NoComm

EPOCH_NUMBER = 1000      # isprs:1642
VAL_INTERVAL = 1000      # isprs:1643
SUPER_BATCH = 100        # isprs:1632


class BatchColors:
    OKBLUE, OKGREEN, WARNING, FAIL, ENDC = "\033[94m", "\033[92m", "\033[93m", "\033[91m", "\033[0m"


def select_best_patch_size(distribution_type, values, patch_acc_loss, patch_occur, is_loss_or_acc="acc",
                           patch_chosen_values=None, debug=False):
    """isprs:549-608 (mutates patch_occur: zeros -> 1, exactly like the reference)."""
    patch_occur[np.where(patch_occur == 0)] = 1
    patch_mean = patch_acc_loss / patch_occur
    if is_loss_or_acc == "acc":
        i = int(np.argmax(patch_mean))
    elif is_loss_or_acc == "loss":
        order = np.argsort(patch_mean)
        i = int([j for j in order if patch_occur[j] > 0][0])
    else:
        raise ValueError("update_type must be acc or loss")
    if patch_chosen_values is not None:
        patch_chosen_values[i] += 1
    cur = int(values[i]) if distribution_type == "multi_fixed" else values[0] + i
    if debug:
        print("patch_acc_loss", patch_acc_loss)
        print("patch_occur", patch_occur)
        print("patch_mean", patch_mean)
        print("Current patch size ", cur)
        if patch_chosen_values is not None:
            print("Distr of chosen sizes ", patch_chosen_values)
    return cur


def _cm_str(cm):
    return np.array_str(np.asarray(cm)).replace("\n", "")


def save_checkpoint(net, output_path, step, patch_acc_loss=None, patch_occur=None, patch_chosen_values=None):
    """saver.save(sess, output_path + 'model', global_step=step) + the three .npy side files (isprs:1798-1802)."""
    np.savez(output_path + "model-" + str(step) + ".npz", **net.state_dict())
    if patch_acc_loss is not None:
        np.save(output_path + "patch_acc_loss_step_" + str(step) + ".npy", patch_acc_loss)
        np.save(output_path + "patch_occur_step_" + str(step) + ".npy", patch_occur)
        np.save(output_path + "patch_chosen_values_step_" + str(step) + ".npy", patch_chosen_values)


def load_checkpoint(net, former_model_path):
    """`saver_restore.restore(sess, former_model_path)` (isprs:1715): a TensorFlow V2 checkpoint written by the reference
    (`<path>.index` + `.data-*`, read by tf_checkpoint.py) or this build's own `<path>.npz`."""
    import os
    if os.path.isfile(former_model_path + ".index"):
        from . import tf_checkpoint
        tf_checkpoint.load_tf_checkpoint(net, former_model_path)
        print(BatchColors.OKBLUE + "Model restored from " + former_model_path + " (TensorFlow bundle)" + BatchColors.ENDC)
        return
    path = former_model_path if former_model_path.endswith(".npz") else former_model_path + ".npz"
    with np.load(path) as d:
        net.load_state_dict({k: d[k] for k in d.files})
    print(BatchColors.OKBLUE + "Model restored from " + former_model_path + BatchColors.ENDC)


def step_from_model_path(former_model_path):
    """isprs:1709: int(former_model_path.split('-')[-1])."""
    return int(former_model_path.replace(".npz", "").split("-")[-1])


# ------------------------------------------------------------------------------------------------- validation
def validation(net, test_pool, selected_testing_instances, mean_full, std_full, batch_size, step, crop_size, comm=None):
    """isprs:1569-1618: forward-only over the held-out instances at `crop_size`, one confusion matrix.
    Returns (confusion matrix, pixels processed)."""
    comm = comm or NoComm()
    K = net.plan.K
    n = len(selected_testing_instances)
    nb = -(-n // batch_size)
    net.conf.zero_()
    for i in range(nb):
        if i % comm.world != comm.rank:          # batches are independent: round-robin over ranks
            continue
        rows = selected_testing_instances[i * batch_size:min((i + 1) * batch_size, n)]
        for j in range(0, len(rows), net.b_max):
            part = rows[j:j + net.b_max]
            P.crop_to_net(net, test_pool, part, crop_size, mean_full, std_full)
            net.forward(len(part), crop_size, want_logits=False, labels=True)
    cm_dev = net.conf.clone()
    comm.all_reduce_sum(cm_dev)
    cm = cm_dev.cpu().numpy().reshape(K, K).astype(np.uint32)
    total, oa, na = MT.overall_and_normalized(cm)
    if comm.rank == 0:
        print("---- Iter " + str(step) +
              " -- Time " + str(datetime.datetime.now().time()) +
              " -- Validation: Overall Accuracy= " + str(total) +
              " Overall Accuracy= " + "{:.6f}".format(oa) +
              " Normalized Accuracy= " + "{:.6f}".format(na) +
              " F1 Score= " + "{:.4f}".format(MT.f1_macro(cm)) +
              " Kappa= " + "{:.4f}".format(MT.cohen_kappa(cm)) +
              " Confusion Matrix= " + _cm_str(cm))
    return cm, n * crop_size * crop_size


def check_training_labels(pool, num_classes, void_label=None):
    """tf.nn.sparse_softmax_cross_entropy_with_logits raises on a label outside [0, K) (isprs:1093); a label map that holds one
    (an unknown colour from datasets.convert_to_class becomes 255 in the uint8 pool) must not silently enter the loss."""
    bad = pool.labels >= num_classes
    if void_label is not None:
        bad &= pool.labels != void_label
    if bool(bad.any()):
        raise ValueError("training labels hold class ids outside [0, %d): %s" % (num_classes, torch.unique(pool.labels[bad]).tolist()[:8]))


# ------------------------------------------------------------------------------------------------- data parallelism
def sync_rng(comm):
    """Data parallelism runs the same host code on every rank and relies on identical `random` / `numpy.random` streams (size
    draw, batch indices, augmentation).  Nothing guarantees that by itself -- the reference never seeds, and a rank that loads
    a cache file skips the draws the rank that built it made -- so at every point where the ranks may have drifted (start of a
    loop, after a cache was built or loaded) rank 0 draws two seeds from ITS streams and every rank re-seeds from them.
    A single process is untouched (the reference's stream order)."""
    if not getattr(comm, "sync_rng", False):
        return
    seeds = comm.broadcast_object((random.getrandbits(31), int(np.random.randint(0, 2 ** 31 - 1))))
    random.seed(seeds[0])
    np.random.seed(seeds[1])


def rank0_call(comm, fn, what):
    """fn() on rank 0 only, its result on every rank.  If rank 0 raises (a bad cache file, an I/O error, a failing builder) EVERY rank
    raises -- the others are not left waiting in the broadcast until the communicator times out."""
    v, err = None, None
    if comm.rank == 0:
        try:
            v = fn()
        except Exception as e:
            if comm.world == 1:
                raise
            err = "%s: %s" % (type(e).__name__, e)
    ok, payload = comm.broadcast_object((err is None, v if err is None else err))
    if not ok:
        raise RuntimeError("rank 0 failed on %s: %s" % (what, payload))
    return payload


def rank0_cached(comm, path, make):
    """the reference's cwd / output .npy caches (isprs:1634-1639, 2087-2115) under data parallelism: rank 0 loads or builds and
    saves, every rank gets rank 0's array (no rank reads a half-written file, every rank takes the same branch)."""
    def load_or_make():
        if os.path.isfile(path):
            return np.load(path, allow_pickle=True)
        v = make()
        tmp = path + ".tmp%d.npy" % os.getpid()
        np.save(tmp, np.asarray(v, dtype=object) if isinstance(v, list) else v)
        os.replace(tmp, path)                    # atomic: a concurrent reader sees the old state or the whole file
        return v
    return rank0_call(comm, load_or_make, path)


# ------------------------------------------------------------------------------------------------- training
class _Pending(object):
    """Results of a step that are read back one step late."""

    def __init__(self, net, out, size_index, step, epoch_counter):
        # pinned destinations: the device-to-host copies are truly asynchronous, the host keeps enqueueing
        self.conf = torch.empty(out["conf"].shape, dtype=out["conf"].dtype, pin_memory=True)
        self.loss_parts = torch.empty(out["loss_parts"].shape, dtype=out["loss_parts"].dtype, pin_memory=True)
        self.conf.copy_(out["conf"], non_blocking=True)
        self.loss_parts.copy_(out["loss_parts"], non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self.size_index, self.step, self.epoch_counter, self.wd = size_index, step, epoch_counter, net.wd

    def get(self):
        self.event.synchronize()
        cm = self.conf.numpy().astype(np.uint32)
        lp = self.loss_parts.numpy()
        return cm, float(lp[0] + self.wd * lp[1])


def train(training_data, training_labels, training_class_distribution, training_rotation_distribution, testing_data,
          testing_labels, testing_class_distribution, testing_instances, lr_initial, batch_size, niter, weight_decay,
          mean_full, std_full, update_type, distribution_type, values, patch_acc_loss, patch_occur, patch_chosen_values,
          probs, resample_batch, output_path, display_step, net_type, dataset, former_model_path=None, *,
          num_classes=6, device="cuda:0", comm=None, noise="device", lr_decay_factor=0.5, tile_dtype=np.float64,
          loss_score_scaled_by_epoch=True, quiet_sizes=False, val_cache_dir=None):
    """isprs:1621-1851, same positional parameters.  Returns the trained DilatedNet."""
    comm = comm or NoComm()
    say = (lambda *a: print(*a)) if comm.rank == 0 else (lambda *a: None)
    say(BatchColors.OKGREEN + "TRAINING" + BatchColors.ENDC)
    channels = training_data[0].shape[-1]
    say("channels ", channels)

    if batch_size % comm.world:
        raise ValueError("batch_size must be divisible by the number of ranks")
    sync_rng(comm)
    selected_training_instances = SP.select_super_batch_instances(training_class_distribution, training_rotation_distribution,
                                                                  batch_size, super_batch=SUPER_BATCH)
    total_length = len(selected_training_instances)
    cache = os.path.join(val_cache_dir or os.getcwd(), "dataset_" + dataset + ".npy")      # isprs:1634-1639
    selected_testing_instances = rank0_cached(comm, cache, lambda: SP.select_super_batch_instances(
        testing_class_distribution, batch_size=batch_size, super_batch=SUPER_BATCH))
    sync_rng(comm)                                   # rank 0 may just have drawn what the others did not
    b_local = batch_size // comm.world
    sl = shard_slice(batch_size, comm.rank, comm.world)
    s_max = int(values[0]) if distribution_type == "single_fixed" else int(max(values))
    net = DilatedNet(net_type, channels, num_classes, weight_decay, b_max=b_local, s_max=s_max, device=device, comm=comm,
                     lr_decay_factor=lr_decay_factor)
    train_pool = P.TilePool(training_data, training_labels, device, dtype=tile_dtype)
    test_pool = P.TilePool(testing_data, testing_labels, device, dtype=tile_dtype)
    check_training_labels(train_pool, num_classes)

    shuffle = np.asarray(random.sample(range(total_length), total_length))
    epoch_counter = 1
    current_iter = 1
    sized = distribution_type in ("multi_fixed", "uniform", "multinomial")
    if former_model_path is not None and "model" in former_model_path:
        current_iter = step_from_model_path(former_model_path)
        if sized:
            patch_acc_loss = np.load(output_path + "patch_acc_loss_step_" + str(current_iter) + ".npy")
            patch_occur = np.load(output_path + "patch_occur_step_" + str(current_iter) + ".npy")
            patch_chosen_values = np.load(output_path + "patch_chosen_values_step_" + str(current_iter) + ".npy")
        load_checkpoint(net, former_model_path)
    else:
        say("Model totally initialized!")

    it = 0
    epoch_mean = 0.0
    epoch_cm_train = np.zeros((num_classes, num_classes), dtype=np.uint32)
    pending = []
    last = dict(cm=None, loss=None, acc=0)

    def consume(p):
        nonlocal epoch_mean, epoch_cm_train
        cm, loss = p.get()
        acc, _, acc_norm = MT.overall_and_normalized(cm)
        epoch_mean += acc
        epoch_cm_train += cm
        if sized:
            if update_type == "loss":
                patch_acc_loss[p.size_index] += loss * (p.epoch_counter / 10.0) if loss_score_scaled_by_epoch else loss
            else:
                patch_acc_loss[p.size_index] += acc_norm
            patch_occur[p.size_index] += 1
        last.update(cm=cm, loss=loss, acc=acc)

    def flush():
        while pending:
            consume(pending.pop(0))

    step = current_iter
    for step in range(current_iter, niter + 1):
        cur_patch_size, cur_size_int = P.draw_patch_size(distribution_type, values, probs)
        if not quiet_sizes:
            say(cur_patch_size)
        shuffle, batch, it = P.select_batch(shuffle, batch_size, it, total_length)
        if step - current_iter < 3:                  # the ranks must move in lock step: same size, same instances
            comm.agree((cur_patch_size, batch[0], batch[-1], it), "patch size / batch indices at step %d" % step)
        rows = selected_training_instances[batch]
        aug = P.draw_augmentation(rows, cur_patch_size, channels, noise=noise)
        mine = P.Augmentation(b_local)
        mine.rot_on, mine.rot, mine.noise_on, mine.flip = aug.rot_on[sl], aug.rot[sl], aug.noise_on[sl], aug.flip[sl]
        mine.noise = aug.noise[sl] if aug.noise is not None else None
        mine.seed, mine.index0 = aug.seed, sl.start      # device noise is keyed by the patch's place in the global batch
        P.crop_to_net(net, train_pool, rows[sl], cur_patch_size, mean_full, std_full, mine)
        out = net.train_step(b_local, cur_patch_size, lr_initial)
        pending.append(_Pending(net, out, cur_size_int, step, epoch_counter))
        while len(pending) > 1:
            consume(pending.pop(0))

        if step != 0 and step % display_step == 0:
            flush()
            cm = last["cm"]
            _, oa, na = MT.overall_and_normalized(cm)
            say("Iter " + str(step) + " -- Time " + str(datetime.datetime.now().time()) +
                " -- Training Minibatch: Loss= " + "{:.6f}".format(last["loss"]) +
                " Absolut Right Pred= " + str(int(last["acc"])) +
                " Overall Accuracy= " + "{:.4f}".format(oa) +
                " Normalized Accuracy= " + "{:.4f}".format(na) +
                " Confusion Matrix= " + _cm_str(cm))

        if step != 0 and step % EPOCH_NUMBER == 0:
            flush()
            _, _, na = MT.overall_and_normalized(epoch_cm_train)
            say("-- Iter " + str(step) + " -- Training Epoch:" +
                " Overall Accuracy= " + "{:.6f}".format(epoch_mean / float(np.sum(epoch_cm_train))) +
                " Normalized Accuracy= " + "{:.6f}".format(na) +
                " Confusion Matrix= " + _cm_str(epoch_cm_train))
            epoch_mean = 0.0
            epoch_cm_train = np.zeros((num_classes, num_classes), dtype=np.uint32)

        if step != 0 and step % VAL_INTERVAL == 0:
            flush()
            if comm.rank == 0:
                save_checkpoint(net, output_path, step, *((patch_acc_loss, patch_occur, patch_chosen_values) if sized else ()))
            cur_patch_val = (select_best_patch_size(distribution_type, values, patch_acc_loss, patch_occur, update_type,
                                                    patch_chosen_values, debug=comm.rank == 0) if sized else int(values[0]))
            validation(net, test_pool, selected_testing_instances, mean_full, std_full, batch_size, step, cur_patch_val, comm)

        if min(it + batch_size, total_length) == total_length or total_length == it + batch_size:     # isprs:1822
            if epoch_counter % resample_batch == 0:
                say("epoch_counter ", epoch_counter)
                selected_training_instances = SP.select_super_batch_instances(training_class_distribution,
                                                                              training_rotation_distribution, batch_size,
                                                                              super_batch=SUPER_BATCH)
                total_length = len(selected_training_instances)
            epoch_counter += 1

    flush()
    say("Optimization Finished!")
    if comm.rank == 0:
        save_checkpoint(net, output_path, step, *((patch_acc_loss, patch_occur, patch_chosen_values) if sized else ()))
    cur_patch_val = (select_best_patch_size(distribution_type, values, patch_acc_loss, patch_occur, update_type,
                                            patch_chosen_values, debug=comm.rank == 0) if sized else int(values[0]))
    validation(net, test_pool, selected_testing_instances, mean_full, std_full, batch_size, step, cur_patch_val, comm)
    return net


# ------------------------------------------------------------------------------------------------- whole tiles
def predict_tile(net, pool, map_index, crop_size, batch_size, mean_full, std_full, comm=None, return_sums=False, flavour="isprs"):
    """The inner loop of validate_test / generate_final_maps (isprs:1261-1284, 1925-1949) for one tile: windows at
    stride floor(s/2) (isprs:1243), logits overlap-added in window order, arg-max of the average.  Returns the
    uint8 label map as a DEVICE tensor [h, w].  Under data parallelism batches of windows go round-robin over the
    ranks and the partial sums are added (sum all-reduce of prob / occur)."""
    from . import _lib
    comm = comm or NoComm()
    h, w = pool.h[map_index], pool.w[map_index]
    K = net.plan.K
    stride = int(math.floor(crop_size / 2.0))
    n_h, n_w = P.window_counts(h, w, crop_size, stride)
    total = n_h * n_w
    if comm.world > 1 and not return_sums and flavour == "isprs" and n_h >= comm.world:
        return _predict_tile_bands(net, pool, map_index, crop_size, batch_size, mean_full, std_full, comm), total
    prob = torch.zeros(h * w * K, dtype=torch.float32, device=net.dev)
    occur = torch.zeros(h * w, dtype=torch.int32, device=net.dev)
    # batches are the REFERENCE's: batch i starts where its `batch_size` puts it (for flavour="contest" that start depends on the
    # batch size, contest:275), whatever this net's b_max is -- under data parallelism b_max is batch_size / world -- and is fed
    # to the net in pieces of at most b_max windows
    nb = -(-total // batch_size)
    st = net._stream()
    for i in range(nb):
        if i % comm.world != comm.rank:
            continue
        pos_all = P.window_positions(h, w, crop_size, stride, i, batch_size, flavour)  # flavour: where batch i starts (patches.window_start)
        f0 = P.window_start(h, w, crop_size, stride, i, batch_size, flavour)
        for c0 in range(0, len(pos_all), net.b_max):
            pos = pos_all[c0:c0 + net.b_max]
            inst = np.concatenate([np.full((len(pos), 1), map_index), pos], axis=1)
            P.crop_to_net(net, pool, inst, crop_size, mean_full, std_full)
            _, logits = net.forward(len(pos), crop_size, want_logits=True)
            _lib.call("drs_stitch_accumulate", prob.data_ptr(), occur.data_ptr(), logits.data_ptr(), h, w, K, crop_size, stride, f0 + c0, len(pos), st)
    if comm.world > 1:          # (the multi-scale caller needs the whole sums; the plain path below exchanges bands instead)
        comm.all_reduce_sum(prob)
        comm.all_reduce_sum(occur)
    if return_sums:
        return prob, occur, total
    out = torch.zeros(h * w, dtype=torch.uint8, device=net.dev)
    _lib.call("drs_stitch_finalize", prob.data_ptr(), occur.data_ptr(), h, w, K, out.data_ptr(), st)
    return out.view(h, w), total


def band_plan(h, crop_size, stride, n_h, world):
    """Window rows per rank and the image rows they touch: rank r takes window rows [a[r], a[r+1]) (contiguous, as even as
    possible); its band is image rows [top[r], bot[r]) (the last window row is shifted back to end at the border, isprs:366-375);
    it OWNS rows [own[r], own[r+1]) of the final map, own[r] = top[r] (own[0] = 0, own[world] = h): what lies below its owned rows
    inside its band is handed to the ranks that own those rows."""
    a = [r * n_h // world for r in range(world + 1)]
    x = lambda i: min(i * stride, h - crop_size)
    top = [x(a[r]) for r in range(world)]
    bot = [x(a[r + 1] - 1) + crop_size for r in range(world)]
    own = [0] + top[1:] + [h]
    return a, top, bot, own


def _predict_tile_bands(net, pool, map_index, crop_size, batch_size, mean_full, std_full, comm):
    """Sliding-window inference of one tile on several ranks: the window rows are cut into one contiguous band per
    rank, every rank overlap-adds its windows into a band-sized accumulator ([rows of the band][w][K] instead of the whole
    [h][w][K]), only the rows a band shares with the next ranks' territory are exchanged (one sum all-reduce of a buffer in which
    every rank fills its own segment: (world-1) x (S - stride) rows instead of the whole map), each rank divides and arg-maxes the
    rows it owns, and the uint8 label bands are gathered.  Sums are formed as (own windows in window order) + (lower ranks'
    contributions in rank order): deterministic, and equal to the single-rank result up to the association of those float sums."""
    from . import _lib
    h, w = pool.h[map_index], pool.w[map_index]
    K = net.plan.K
    S = crop_size
    stride = int(math.floor(S / 2.0))
    n_h, n_w = P.window_counts(h, w, S, stride)
    W, r = comm.world, comm.rank
    a, top, bot, own = band_plan(h, S, stride, n_h, W)
    rows = bot[r] - top[r]
    prob = torch.zeros(rows * w * K, dtype=torch.float32, device=net.dev)
    occur = torch.zeros(rows * w, dtype=torch.int32, device=net.dev)
    st = net._stream()
    bs = min(batch_size, net.b_max)
    f_end = a[r + 1] * n_w
    # the stitch kernel addresses absolute image rows: hand it the address row 0 would have
    vprob, voccur = prob.data_ptr() - top[r] * w * K * 4, occur.data_ptr() - top[r] * w * 4
    for f0 in range(a[r] * n_w, f_end, bs):
        f = np.arange(f0, min(f0 + bs, f_end))
        pos = np.stack([np.minimum((f // n_w) * stride, h - S), np.minimum((f % n_w) * stride, w - S)], axis=1).astype(np.int64)
        inst = np.concatenate([np.full((len(pos), 1), map_index), pos], axis=1)
        P.crop_to_net(net, pool, inst, S, mean_full, std_full)
        _, logits = net.forward(len(pos), S, want_logits=True)
        _lib.call("drs_stitch_accumulate", vprob, voccur, logits.data_ptr(), h, w, K, S, stride, int(f0), len(pos), st)
    # exchange: segment q of the buffer = rank q's band rows below its owned rows, [own[q+1], bot[q])
    seg = [max(0, bot[q] - own[q + 1]) for q in range(W)]
    off = np.concatenate([[0], np.cumsum(seg)]).astype(np.int64)
    xp = torch.zeros(int(off[-1]) * w * K, dtype=torch.float32, device=net.dev)
    xo = torch.zeros(int(off[-1]) * w, dtype=torch.int32, device=net.dev)
    if seg[r]:
        lo = own[r + 1] - top[r]
        xp[off[r] * w * K:off[r + 1] * w * K].copy_(prob[lo * w * K:(lo + seg[r]) * w * K])
        xo[off[r] * w:off[r + 1] * w].copy_(occur[lo * w:(lo + seg[r]) * w])
    comm.all_reduce_sum(xp)
    comm.all_reduce_sum(xo)
    for q in range(r):          # lower ranks' contributions to the rows this rank owns, in rank order
        lo, hi = max(own[q + 1], own[r]), min(bot[q], own[r + 1])
        if hi > lo:
            src, dst = off[q] + (lo - own[q + 1]), lo - top[r]
            prob[dst * w * K:(dst + hi - lo) * w * K] += xp[src * w * K:(src + hi - lo) * w * K]
            occur[dst * w:(dst + hi - lo) * w] += xo[src * w:(src + hi - lo) * w]
    out = torch.zeros(h * w, dtype=torch.uint8, device=net.dev)
    n_own, d0 = own[r + 1] - own[r], own[r] - top[r]
    _lib.call("drs_stitch_finalize", prob.data_ptr() + d0 * w * K * 4, occur.data_ptr() + d0 * w * 4, n_own, w, K, out.data_ptr() + own[r] * w, st)
    comm.all_reduce_sum(out)    # every rank wrote only the rows it owns: the sum is the gather of the uint8 label bands
    return out.view(h, w)


def predict_tile_multiscale(net, pool, map_index, crop_sizes, batch_size, mean_full, std_full, comm=None):
    """isprs:1347-1474 inner part: for every scale the averaged-logit map, softmax over classes, summed; arg-max."""
    from . import _lib
    h, w = pool.h[map_index], pool.w[map_index]
    K = net.plan.K
    acc = torch.zeros(h * w * K, dtype=torch.float32, device=net.dev)
    for s_ in crop_sizes:
        prob, occur, _ = predict_tile(net, pool, map_index, int(s_), batch_size, mean_full, std_full, comm, return_sums=True)
        _lib.call("drs_softmax_accumulate", prob.data_ptr(), occur.data_ptr(), h, w, K, acc.data_ptr(), net._stream())
    ones = torch.ones(h * w, dtype=torch.int32, device=net.dev)
    out = torch.zeros(h * w, dtype=torch.uint8, device=net.dev)
    _lib.call("drs_stitch_finalize", acc.data_ptr(), ones.data_ptr(), h, w, K, out.data_ptr(), net._stream())
    return out.view(h, w)


def best_sizes(distribution_type, values, patch_acc_loss, patch_occur, update_type, num_scales):
    """The reference picks the best size, removes it from the candidates and repeats (isprs:1370-1420)."""
    values = np.asarray(values).copy()
    acc, occ = np.asarray(patch_acc_loss).copy(), np.asarray(patch_occur).copy()
    chosen = []
    for _ in range(num_scales):
        if distribution_type not in ("multi_fixed", "uniform", "multinomial"):
            chosen.append(int(values[0]))
            continue
        crop = select_best_patch_size(distribution_type, values, acc, occ, update_type)
        chosen.append(int(crop))
        ind = np.where(values == crop)
        values, acc, occ = np.delete(values, ind), np.delete(acc, ind), np.delete(occ, ind)
    return chosen


def validate_test(net, testing_data, testing_labels, testing_instances, batch_size, mean_full, std_full, crop_size, step,
                  output_path=None, comm=None, pool=None, ignore_label=6, crop_sizes=None, flavour="isprs"):
    """isprs:1241-1344: per tile, sliding-window prediction and scores (label 6 = eroded boundary is skipped,
    isprs:1294).  Returns (all-maps confusion matrix, list of label maps as numpy)."""
    from . import _lib
    comm = comm or NoComm()
    K = net.plan.K
    pool = pool or P.TilePool(testing_data, testing_labels, net.dev)
    all_cm = np.zeros((K, K), dtype=np.uint32)
    all_kappa = np.zeros(len(testing_data), dtype=np.float32)
    all_f1 = np.zeros(len(testing_data), dtype=np.float32)
    all_f1_per_class = np.zeros(K, dtype=np.float32)
    maps = []
    for k in range(len(testing_data)):
        if crop_sizes:      # validate_test_multiscale (isprs:1347-1474): several sizes, softmax maps summed
            pred = predict_tile_multiscale(net, pool, k, crop_sizes, batch_size, mean_full, std_full, comm)
        else:
            pred, _ = predict_tile(net, pool, k, crop_size, batch_size, mean_full, std_full, comm, flavour=flavour)
        h, w = pool.h[k], pool.w[k]
        conf = torch.zeros(K * K, dtype=torch.int32, device=net.dev)
        lab = pool.labels[int(pool.lab_off[k].item()):int(pool.lab_off[k].item()) + h * w]
        _lib.call("drs_confusion", lab.data_ptr(), pred.data_ptr(), None, h * w, K, ignore_label, conf.data_ptr(), net._stream())
        cm = conf.cpu().numpy().reshape(K, K).astype(np.uint32)
        all_cm += cm
        total, oa, na = MT.overall_and_normalized(cm)
        f1c, present = MT.f1_per_class(cm)
        f1_full = np.zeros(K, dtype=np.float32)
        f1_full[present] = f1c
        all_kappa[k], all_f1[k] = MT.cohen_kappa(cm), MT.f1_macro(cm)
        all_f1_per_class += f1_full
        maps.append(pred.cpu().numpy())
        if comm.rank == 0:
            print("---- Iter " + str(step) +
                  " -- Test Map " + str(testing_instances[k]) + ": Overall Accuracy= " + str(total) +
                  " Overall Accuracy= " + "{:.6f}".format(oa) +
                  " Normalized Accuracy= " + "{:.6f}".format(na) +
                  " F1 Score per class= " + np.array_str(f1_full).replace("\n", "") +
                  " F1 Score= " + "{:.4f}".format(all_f1[k]) +
                  " Kappa= " + "{:.4f}".format(all_kappa[k]) +
                  " Confusion Matrix= " + _cm_str(cm))
    total, oa, na = MT.overall_and_normalized(all_cm)
    if comm.rank == 0:
        print("---- Iter " + str(step) +
              " -- Test ALL MAPS: Overall Accuracy= " + str(total) +
              " Overall Accuracy= " + "{:.6f}".format(oa) +
              " Normalized Accuracy= " + "{:.6f}".format(na) +
              " F1 Score= " + np.array_str(all_f1).replace("\n", " ") +
              " Mean F1 Score= " + "{:.6f}".format(np.sum(all_f1) / float(len(testing_data))) +
              " F1 Score per class= " + np.array_str(all_f1_per_class / float(len(testing_data))).replace("\n", "") +
              " Kappa= " + np.array_str(all_kappa).replace("\n", " ") +
              " Mean Kappa Score= " + "{:.6f}".format(np.sum(all_kappa) / float(len(testing_data))) +
              " Confusion Matrix= " + _cm_str(all_cm))
    return all_cm, maps


def generate_final_maps(net, testing_data, testing_instances, batch_size, mean_full, std_full, update_type,
                        distribution_type, values, dataset, output_path, patch_acc_loss=None, patch_occur=None, comm=None):
    """isprs:1854-1957: best (or fixed) patch size, sliding-window label map per tile, written as the reference's colour TIFF
    (`top_mosaic_09cm_area<i>_class.tif` / `top_potsdam_<i>_label.tif`) and as class ids (`.npy`)."""
    comm = comm or NoComm()
    sized = distribution_type in ("multi_fixed", "uniform", "multinomial")
    crop_size = (select_best_patch_size(distribution_type, values, patch_acc_loss, patch_occur, update_type, debug=comm.rank == 0)
                 if sized else int(values[0]))
    pool = P.TilePool(testing_data, None, net.dev)
    maps = []
    for k in range(len(testing_data)):
        pred, _ = predict_tile(net, pool, k, crop_size, batch_size, mean_full, std_full, comm)
        maps.append(pred.cpu().numpy())
        if comm.rank == 0 and output_path:
            # isprs:1950-1955: the colour map under the reference's file names (ISPRS palette, isprs:118-139), plus the class ids as .npy
            from . import datasets
            stem = ("top_mosaic_09cm_area" + str(testing_instances[k]) + "_class" if dataset == "vaihingen"
                    else "top_potsdam_" + str(testing_instances[k]) + "_label")
            datasets.create_prediction_map(output_path + stem + ".tif", maps[-1])
            np.save(output_path + stem + ".npy", maps[-1])
    return maps
