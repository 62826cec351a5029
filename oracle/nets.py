"""Layer tables of the four BASELINE nets, restated as data.

Follows /root/reference/isprs_dilated_random.py:
  dilated_icpr_original      (Dilated6)         :761-788
  dilated_icpr_rate6_densely (DenseDilated6)    :914-959
  dilated_grsl               (Dilated6Pooling)  :962-993
  dilated_grsl_rate8         (Dilated8Pooling)  :996-1033   (net_type 'dilated8_grsl' at :1672)

Each conv entry is (scope, k, c_in, c_out, rate); c_in == -1 means "input channels".
"""

NETS = {
    "dilated_icpr_original": dict(
        act="relu", pool=False, dense=False,
        convs=[("main_conv1", 5, -1, 64, 1), ("main_conv2", 5, 64, 64, 1),
               ("main_conv3", 4, 64, 128, 2), ("main_conv4", 4, 128, 128, 2),
               ("main_conv5", 3, 128, 256, 4), ("main_conv6", 3, 256, 256, 4)],
        c_last=256),
    "dilated_grsl": dict(
        act="lrelu", pool=True, dense=False,
        convs=[("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2),
               ("conv3", 4, 64, 128, 3), ("conv4", 4, 128, 128, 4),
               ("conv5", 3, 128, 256, 5), ("conv6", 3, 256, 256, 6)],
        c_last=256),
    "dilated_grsl_rate8": dict(
        act="lrelu", pool=True, dense=False,
        convs=[("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2),
               ("conv3", 4, 64, 128, 3), ("conv4", 4, 128, 128, 4),
               ("conv5", 3, 128, 192, 5), ("conv6", 3, 192, 192, 6),
               ("conv7", 3, 192, 256, 7), ("conv8", 3, 256, 256, 8)],
        c_last=256),
    # plain-chain variants (isprs:791-815, 857-911; coffee_dilated_random.py:665-707, 788-830)
    "dilated_icpr_rate6": dict(
        act="relu", pool=False, dense=False,
        convs=[("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 128, 3),
               ("conv4", 4, 128, 128, 4), ("conv5", 3, 128, 256, 5), ("conv6", 3, 256, 256, 6)], c_last=256),
    "dilated_icpr_rate6_small": dict(
        act="relu", pool=False, dense=False,
        convs=[("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 64, 3),
               ("conv4", 4, 64, 128, 4), ("conv5", 3, 128, 128, 5), ("conv6", 3, 128, 128, 6)], c_last=128),
    "dilated_icpr_rate6_nodilation": dict(
        act="relu", pool=False, dense=False,
        convs=[("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 1), ("conv3", 4, 64, 128, 1),
               ("conv4", 4, 128, 128, 1), ("conv5", 3, 128, 256, 1), ("conv6", 3, 256, 256, 1)], c_last=256),
    "dilated_icpr_rate1": dict(
        act="relu", pool=False, dense=False,
        convs=[("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 1), ("conv3", 4, 64, 128, 1),
               ("conv4", 4, 128, 128, 1), ("conv5", 3, 128, 256, 1), ("conv6", 3, 256, 256, 1)], c_last=256),
    "dilated_icpr_vary_rate": dict(
        act="relu", pool=False, dense=False,
        convs=[("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 128, 4),
               ("conv4", 4, 128, 128, 1), ("conv5", 3, 128, 256, 2), ("conv6", 3, 256, 256, 4)], c_last=256),
    # contest_dilated_random.py:574-601
    "dilated_icpr_old": dict(
        act="relu", pool=False, dense=False,
        convs=[("conv1", 5, -1, 64, 1), ("conv3", 4, 64, 128, 2), ("conv5", 3, 128, 256, 4)], c_last=256),
    # isprs:818-854 / coffee:721-740: average pools (5,5,5,7,7) after conv1..5, none after conv6
    "dilated_icpr_rate6_avgpool": dict(
        act="relu", pool=False, dense=False, pools=[5, 5, 5, 7, 7, 0],
        convs=[("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 128, 3),
               ("conv4", 4, 128, 128, 4), ("conv5", 3, 128, 256, 5), ("conv6", 3, 256, 256, 6)], c_last=256),
    # isprs:1064-1086 with _squeeze_conv_layer :726-742.  squeezes: (scope, k, in_dim, out_dim, rate, k_dim)
    "dilated_icpr_rate6_squeeze": dict(
        act="relu", pool=False, dense=False, convs=[("conv1", 5, -1, 64, 1)],
        squeezes=[("conv2", 5, 64, 64, 2, 32), ("conv3", 4, 64, 128, 3, 64), ("conv4", 4, 128, 128, 4, 64),
                  ("conv5", 3, 128, 256, 5, 64), ("conv6", 3, 256, 256, 6, 128)], c_last=256),
    # isprs:1036-1061: SE blocks (ratio 4) after conv2 / conv4 / conv6; se: block index -> scope
    "dilated_icpr_rate6_SE": dict(
        act="relu", pool=False, dense=False, se={1: "se1", 3: "se2", 5: "se3"},
        convs=[("conv1", 5, -1, 64, 1), ("conv2", 5, 64, 64, 2), ("conv3", 4, 64, 128, 3),
               ("conv4", 4, 128, 128, 4), ("conv5", 3, 128, 256, 5), ("conv6", 3, 256, 256, 6)], c_last=256),
    "dilated_icpr_rate6_densely": dict(
        act="relu", pool=False, dense=True,
        convs=[("conv1", 5, -1, 32, 1), ("conv2", 5, 32, 32, 2),
               ("conv3", 4, 64, 64, 3), ("conv4", 4, 128, 64, 4),
               ("conv5", 3, 192, 128, 5), ("conv6", 3, 320, 128, 6)],
        c_last=448),
}
# isprs:1672 spells Dilated8Pooling 'dilated8_grsl'; coffee/contest/README spell it 'dilated_grsl_rate8'
ALIASES = {"dilated8_grsl": "dilated_grsl_rate8", "dilated_grsl_old": "dilated_grsl"}


def resolve(net_type):
    return ALIASES.get(net_type, net_type)


def conv_specs(net_type, channels):
    """every conv block (scope, k, c_in, c_out, rate) in execution order; a squeeze layer contributes three
    (`_s1` 1x1 squeeze, `_s2_1` 1x1 expand, `_s2_2` k x k expand: isprs:729-735)."""
    spec = NETS[resolve(net_type)]
    out = [(n, k, channels if ci < 0 else ci, co, r) for (n, k, ci, co, r) in spec["convs"]]
    for (n, k, ind, outd, r, kd) in spec.get("squeezes", []):
        out += [(n + "_s1", 1, ind, kd, r), (n + "_s2_1", 1, kd, outd // 2, r), (n + "_s2_2", k, kd, outd // 2, r)]
    return out


def same_pad(k, rate):
    """tf.nn.atrous_conv2d / conv2d SAME padding, stride 1 (isprs:710-712)."""
    k_eff = k + (k - 1) * (rate - 1)
    total = k_eff - 1
    before = total // 2
    return before, total - before
