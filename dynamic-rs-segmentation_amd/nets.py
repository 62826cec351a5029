"""Net tables and the execution plan of the four BASELINE nets.

Host-side mirror of the reference's net builders (/root/reference/isprs_dilated_random.py:
dilated_icpr_original :761-788, dilated_icpr_rate6_densely :914-959, dilated_grsl :962-993,
dilated_grsl_rate8 :996-1033; dispatch by net_type :1660-1680).  The reference builds a
TensorFlow graph; here a net is a list of `Layer` records that the HIP kernels are driven from.
"""
from collections import namedtuple

# (scope, k, c_in (-1 = image bands), c_out, rate)
_TABLES = {
    "dilated_icpr_original": ("relu", False, False, 256, [
        ("main_conv1", 5, -1, 64, 1), ("main_conv2", 5, 64, 64, 1), ("main_conv3", 4, 64, 128, 2),
        ("main_conv4", 4, 128, 128, 2), ("main_conv5", 3, 128, 256, 4), ("main_conv6", 3, 256, 256, 4)]),
    "dilated_grsl": ("lrelu", True, False, 256, [
        ("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 128, 3),
        ("conv4", 4, 128, 128, 4), ("conv5", 3, 128, 256, 5), ("conv6", 3, 256, 256, 6)]),
    "dilated_grsl_rate8": ("lrelu", True, False, 256, [
        ("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 128, 3), ("conv4", 4, 128, 128, 4),
        ("conv5", 3, 128, 192, 5), ("conv6", 3, 192, 192, 6), ("conv7", 3, 192, 256, 7), ("conv8", 3, 256, 256, 8)]),
    # plain-chain variants outside BASELINE.json's configs: same block, other tables
    "dilated_icpr_rate6": ("relu", False, False, 256, [                       # isprs:890-911, coffee:693-707
        ("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 128, 3),
        ("conv4", 4, 128, 128, 4), ("conv5", 3, 128, 256, 5), ("conv6", 3, 256, 256, 6)]),
    "dilated_icpr_rate6_small": ("relu", False, False, 128, [                 # isprs:791-815, coffee:665-679
        ("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 64, 3),
        ("conv4", 4, 64, 128, 4), ("conv5", 3, 128, 128, 5), ("conv6", 3, 128, 128, 6)]),
    "dilated_icpr_rate6_nodilation": ("relu", False, False, 256, [            # isprs:857-887 (is_normal_conv=True)
        ("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 1), ("conv3", 4, 64, 128, 1),
        ("conv4", 4, 128, 128, 1), ("conv5", 3, 128, 256, 1), ("conv6", 3, 256, 256, 1)]),
    "dilated_icpr_rate1": ("relu", False, False, 256, [                       # coffee:788-802
        ("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 1), ("conv3", 4, 64, 128, 1),
        ("conv4", 4, 128, 128, 1), ("conv5", 3, 128, 256, 1), ("conv6", 3, 256, 256, 1)]),
    "dilated_icpr_vary_rate": ("relu", False, False, 256, [                   # coffee:816-830
        ("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 128, 4),
        ("conv4", 4, 128, 128, 1), ("conv5", 3, 128, 256, 2), ("conv6", 3, 256, 256, 4)]),
    # contest's shallow net (contest_dilated_random.py:574-601): conv1, conv3, conv5 only
    "dilated_icpr_old": ("relu", False, False, 256, [
        ("conv1", 5, -1, 64, 1), ("conv3", 4, 64, 128, 2), ("conv5", 3, 128, 256, 4)]),
    # average-pooled variant (isprs:818-854, coffee:721-740): 5x5 / 7x7 stride-1 SAME average pools, none after conv6
    "dilated_icpr_rate6_avgpool": ("relu", [("avg", 5), ("avg", 5), ("avg", 5), ("avg", 7), ("avg", 7), None], False, 256, [
        ("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 128, 3),
        ("conv4", 4, 128, 128, 4), ("conv5", 3, 128, 256, 5), ("conv6", 3, 256, 256, 6)]),
    # squeeze variant (isprs:1064-1086 with _squeeze_conv_layer :726-742): conv1, then per layer a 1x1 squeeze to k_dim and
    # two parallel expands (1x1 and k x k dilated, out_dim/2 channels each) whose outputs are concatenated.
    # entries: (scope, k, in_dim, out_dim, rate, k_dim)
    "dilated_icpr_rate6_squeeze": ("relu", False, "squeeze", 256, [
        ("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2, 32), ("conv3", 4, 64, 128, 3, 64),
        ("conv4", 4, 128, 128, 4, 64), ("conv5", 3, 128, 256, 5, 64), ("conv6", 3, 256, 256, 6, 128)]),
    # squeeze-and-excitation variant (isprs:1036-1061): rate6 with an SE block (ratio 4) after conv2, conv4 and conv6
    "dilated_icpr_rate6_SE": ("relu", False, False, 256, [
        ("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 128, 3),
        ("conv4", 4, 128, 128, 4), ("conv5", 3, 128, 256, 5), ("conv6", 3, 256, 256, 6)]),
    "dilated_icpr_rate6_densely": ("relu", False, True, 448, [
        ("conv1", 5, -1, 32, 1), ("conv2", 5, 32, 32, 2), ("conv3", 4, 64, 64, 3),
        ("conv4", 4, 128, 64, 4), ("conv5", 3, 192, 128, 5), ("conv6", 3, 320, 128, 6)]),
}
# isprs:1672 spells Dilated8Pooling 'dilated8_grsl'; coffee:1203, contest:1004 and README:33 'dilated_grsl_rate8'
# contest's 'dilated_grsl_old' (contest:604-641) is layer for layer dilated_grsl
_ALIASES = {"dilated8_grsl": "dilated_grsl_rate8", "dilated_grsl_old": "dilated_grsl"}
# block index -> SE scope placed after that block (isprs:1042, 1046, 1050); ratio 4
_SE = {"dilated_icpr_rate6_SE": {1: "se1", 3: "se2", 5: "se3"}}
SE_RATIO = 4

# one conv block (conv + bias + BN + activation [+ pool]); it reads channels [0, cin) of activation buffer `src` and writes its
# output into channels [dst_coff, dst_coff + cout) of buffer `dst`
Layer = namedtuple("Layer", "name k cin cin_k cout rate pad_b pad_a halo src dst dst_coff")


def known_net_types():
    return sorted(list(_TABLES) + list(_ALIASES))


def resolve(net_type):
    name = _ALIASES.get(net_type, net_type)
    if name not in _TABLES:
        # the reference prints a red message and returns None (isprs:1679-1680); the mirror raises
        raise ValueError("Error! Net type not identified: " + str(net_type))
    return name


def same_pad(k, rate):
    """TF SAME padding at stride 1: total (k-1)*rate, the extra pixel goes after (bottom/right)."""
    total = (k - 1) * rate
    return total // 2, total - total // 2


def round_up(v, m):
    return (v + m - 1) // m * m


class Plan(object):
    """Static description of one net for given (bands, classes)."""

    def __init__(self, net_type, channels, num_classes, first_cin_pad=8):
        """first_cin_pad: channel count the image bands are padded to in the conv1 input slab: 8 (several filter taps share a
        32-deep K-step of the fp32 kernels) or 32 (one tap per K-step; what the split-bf16 kernels take)."""
        self.net_type = resolve(net_type)
        act, pool, topo, self.c_last, convs = _TABLES[self.net_type]
        self.dense = topo is True
        self.alpha = 0.0 if act == "relu" else 0.1      # max(alpha*x, x): ReLU / leaky ReLU (isprs:620-621)
        self.channels = channels
        self.K = num_classes
        # ---- wiring: blocks in execution order + the activation buffers they read / write (name -> [channels, halo]).
        # A chain gives every block its own output slab; the dense net (isprs:921-948) and the squeeze net (isprs:737-742)
        # concatenate by writing channel slices of a shared slab, so no concat copy exists.
        blocks = []        # (name, k, cin, cout, rate, src, dst, dst_coff)
        chan = {"x0": round_up(channels, first_cin_pad if channels <= first_cin_pad else 32)}
        if topo == "squeeze":
            name, k, _, co, r = convs[0]
            blocks.append((name, k, channels, co, r, "x0", "c1", 0))
            chan["c1"] = co
            for j, (name, k, ind, outd, r, kd) in enumerate(convs[1:], start=2):
                a, c, prev = "a%d" % j, "c%d" % j, "c%d" % (j - 1)
                chan[a], chan[c] = kd, outd
                blocks.append((name + "_s1", 1, ind, kd, r, prev, a, 0))
                blocks.append((name + "_s2_1", 1, kd, outd // 2, r, a, c, 0))
                blocks.append((name + "_s2_2", k, kd, outd // 2, r, a, c, outd // 2))
            self.feat = "c%d" % len(convs)
        elif self.dense:
            off = 0
            chan["concat"] = self.c_last
            for i, (name, k, ci, co, r) in enumerate(convs):
                blocks.append((name, k, channels if ci < 0 else ci, co, r, "x0" if i == 0 else "concat", "concat", off))
                off += co
            self.feat = "concat"
        else:
            for i, (name, k, ci, co, r) in enumerate(convs):
                dst = "x%d" % (i + 1) if i + 1 < len(convs) else "feat"
                blocks.append((name, k, channels if ci < 0 else ci, co, r, "x%d" % i, dst, 0))
                chan[dst] = co
            self.feat = "feat"
        # per-block pooling after the activation: None | ("max", 3) | ("avg", k)
        self.pools = list(pool) if isinstance(pool, list) else [("max", 3) if pool else None] * len(blocks)
        self.pool = any(q is not None and q[0] == "max" for q in self.pools)
        self.layers = []
        halo = {n: 0 for n in chan}
        for (name, k, ci, co, r, src, dst, coff) in blocks:
            pb, pa = same_pad(k, r)
            halo[src] = max(halo[src], pb, pa)          # a slab's halo must cover every conv that reads it
            cin_k = chan["x0"] if src == "x0" else round_up(ci, 32)
            self.layers.append(Layer(name, k, ci, cin_k, co, r, pb, pa, max(pb, pa), src, dst, coff))
        self.buffers = {n: (chan[n], halo[n]) for n in chan}
        # flat parameter layout: every kernel (HWIO), then every bias; the classifier last in both groups
        self.offsets = {}
        off = 0
        for L in self.layers:
            self.offsets[L.name + "/weights"] = (off, (L.k, L.k, L.cin, L.cout))
            off += L.k * L.k * L.cin * L.cout
        self.offsets["conv_classifier/weights"] = (off, (1, 1, self.c_last, num_classes))
        off += self.c_last * num_classes
        # squeeze-and-excitation FC layers (_fc_layer isprs:666-679: weights decayed, biases 0.1)
        self.se = dict(_SE.get(self.net_type, {}))
        for i, scope in sorted(self.se.items()):
            C = self.layers[i].cout
            for fc, shape in (("_fc1", (C, C // SE_RATIO)), ("_fc2", (C // SE_RATIO, C))):
                self.offsets[scope + fc + "/weights"] = (off, shape)
                off += shape[0] * shape[1]
        self.n_decay = off                              # weight decay applies to kernels only (isprs:640-652)
        for L in self.layers:
            self.offsets[L.name + "/biases"] = (off, (L.cout,))
            off += L.cout
        self.offsets["conv_classifier/biases"] = (off, (num_classes,))
        off += num_classes
        for i, scope in sorted(self.se.items()):
            C = self.layers[i].cout
            for fc, n in (("_fc1", C // SE_RATIO), ("_fc2", C)):
                self.offsets[scope + fc + "/biases"] = (off, (n,))
                off += n
        self.n_params = off
        # batch-norm moving statistics: per block mean[C] then variance[C]
        self.bn_offsets = {}
        off = 0
        for L in self.layers:
            self.bn_offsets[L.name] = off
            off += 2 * L.cout
        self.n_bn = off
        if self.dense:      # kept for callers that want the slice table (isprs:921-948)
            self.concat_off = [L.dst_coff for L in self.layers]
            self.concat_halo = self.buffers["concat"][1]

    def mac_per_pixel(self):
        return sum(L.k * L.k * L.cin * L.cout for L in self.layers) + self.c_last * self.K
