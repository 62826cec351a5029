"""Independent second implementation of the reference graph on PyTorch-CPU autograd.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Two uses only:
  * cross-check of oracle/tf_ops.py (different conv code, autograd instead of the
    hand-derived backward, torch's own batch-norm / max-pool / cross-entropy / SGD);
  * the ``cpu_baseline`` leg of bench.py (kind "port"): the reference step expressed
    with CPU fp32 ops, timed on the GPU box's host cores.

Graph restated from /root/reference/isprs_dilated_random.py:700-723 (block),
:761-1033 (nets), :1089-1099 (loss), :1685-1687 (optimizer).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import nets as _nets
from .tf_ops import BN_EPS, BN_DECAY, MOMENTUM, learning_rate


class TorchNet(object):
    def __init__(self, net_type, channels, num_classes, params=None, dtype=torch.float32):
        self.spec = _nets.NETS[_nets.resolve(net_type)]
        self.convs = _nets.conv_specs(net_type, channels)
        self.K = num_classes
        self.dtype = dtype
        self.w, self.b, self.mm, self.mv = {}, {}, {}, {}
        for (name, k, ci, co, r) in self.convs:
            # stored OIHW for F.conv2d; set_params converts from the reference's HWIO
            self.w[name] = torch.zeros(co, ci, k, k, dtype=dtype, requires_grad=True)
            self.b[name] = torch.full((co,), 0.1, dtype=dtype, requires_grad=True)
            self.mm[name] = torch.zeros(co, dtype=dtype)
            self.mv[name] = torch.ones(co, dtype=dtype)
        self.fcw, self.fcb = {}, {}
        for li, scope in sorted(self.spec.get("se", {}).items()):
            C = self.convs[li][3]
            for fc, (a, b_) in (("_fc1", (C, C // 4)), ("_fc2", (C // 4, C))):
                self.fcw[scope + fc] = torch.zeros(a, b_, dtype=dtype, requires_grad=True)
                self.fcb[scope + fc] = torch.zeros(b_, dtype=dtype, requires_grad=True)
        cl = self.spec["c_last"]
        self.w["conv_classifier"] = torch.zeros(num_classes, cl, 1, 1, dtype=dtype, requires_grad=True)
        self.b["conv_classifier"] = torch.zeros(num_classes, dtype=dtype, requires_grad=True)
        self.global_step = 0
        if params is not None:
            self.set_params(params)
        self._opt = None

    def set_params(self, params):
        with torch.no_grad():
            for n in self.w:
                self.w[n].copy_(torch.from_numpy(np.ascontiguousarray(
                    np.transpose(params[n + "/weights"], (3, 2, 0, 1)))).to(self.dtype))
                self.b[n].copy_(torch.from_numpy(np.asarray(params[n + "/biases"])).to(self.dtype))
            for n in self.fcw:
                self.fcw[n].copy_(torch.from_numpy(np.asarray(params[n + "/weights"])).to(self.dtype))
                self.fcb[n].copy_(torch.from_numpy(np.asarray(params[n + "/biases"])).to(self.dtype))
            for n in self.mm:
                self.mm[n].copy_(torch.from_numpy(np.asarray(params[n + "/moving_mean"])).to(self.dtype))
                self.mv[n].copy_(torch.from_numpy(np.asarray(params[n + "/moving_variance"])).to(self.dtype))

    def get_params(self):
        out = {}
        for n in self.w:
            out[n + "/weights"] = np.transpose(self.w[n].detach().numpy(), (2, 3, 1, 0)).copy()
            out[n + "/biases"] = self.b[n].detach().numpy().copy()
        for n in self.fcw:
            out[n + "/weights"] = self.fcw[n].detach().numpy().copy()
            out[n + "/biases"] = self.fcb[n].detach().numpy().copy()
        for n in self.mm:
            out[n + "/moving_mean"] = self.mm[n].numpy().copy()
            out[n + "/moving_variance"] = self.mv[n].numpy().copy()
        return out

    def _block(self, li, inp, is_training):
        name, k, ci, co, r = self.convs[li]
        pb, pa = _nets.same_pad(k, r)
        z = F.conv2d(F.pad(inp, (pb, pa, pb, pa)), self.w[name], self.b[name], dilation=r)
        # torch's running_var update uses the unbiased batch variance, as TF's fused kernel does
        y = F.batch_norm(z, self.mm[name], self.mv[name], None, None, training=is_training,
                         momentum=1.0 - BN_DECAY, eps=BN_EPS)
        y = F.relu(y) if self.spec["act"] == "relu" else torch.maximum(0.1 * y, y)
        ak = self.spec.get("pools", [0] * len(self.convs))[li]
        if self.spec["pool"]:
            y = F.max_pool2d(y, 3, 1, 1)
        elif ak:
            y = F.avg_pool2d(y, ak, 1, ak // 2, count_include_pad=False)
        return y

    def forward(self, x_nhwc, is_training):
        x = torch.as_tensor(x_nhwc, dtype=self.dtype).permute(0, 3, 1, 2)
        n = len(self.convs)
        if self.spec.get("squeezes"):
            cur = self._block(0, x, is_training)
            for li in range(1, n, 3):
                s1 = self._block(li, cur, is_training)
                cur = torch.cat([self._block(li + 1, s1, is_training), self._block(li + 2, s1, is_training)], dim=1)
        elif self.spec["dense"]:
            cur = self._block(0, x, is_training)
            for li in range(1, n):
                cur = torch.cat([cur, self._block(li, cur, is_training)], dim=1)
        else:
            cur = x
            for li in range(n):
                cur = self._block(li, cur, is_training)
                if li in self.spec.get("se", {}):                           # isprs:682-697
                    sc = self.spec["se"][li]
                    sq = cur.mean(dim=(2, 3))
                    ex = torch.sigmoid(F.relu(sq @ self.fcw[sc + "_fc1"] + self.fcb[sc + "_fc1"]) @ self.fcw[sc + "_fc2"] + self.fcb[sc + "_fc2"])
                    cur = cur * ex[:, :, None, None]
        logits = F.conv2d(cur, self.w["conv_classifier"], self.b["conv_classifier"])
        return logits.permute(0, 2, 3, 1)

    def loss(self, logits, y, weight_decay, mask=None):
        lg = logits.reshape(-1, self.K)
        yy = torch.as_tensor(np.asarray(y).reshape(-1), dtype=torch.long)
        if mask is not None:
            m = torch.as_tensor(np.asarray(mask).reshape(-1).astype(bool))
            lg, yy = lg[m], yy[m]
        ce = F.cross_entropy(lg, yy, reduction="mean")
        l2 = sum(0.5 * (w ** 2).sum() for w in list(self.w.values()) + list(self.fcw.values()))
        return ce + weight_decay * l2

    def params_list(self):
        return list(self.w.values()) + list(self.b.values()) + list(self.fcw.values()) + list(self.fcb.values())

    def grads(self, x, y, weight_decay, mask=None):
        for p in self.params_list():
            p.grad = None
        logits = self.forward(x, True)
        loss = self.loss(logits, y, weight_decay, mask)
        loss.backward()
        g = {}
        for n in self.w:
            g[n + "/weights"] = np.transpose(self.w[n].grad.numpy(), (2, 3, 1, 0)).copy()
            g[n + "/biases"] = self.b[n].grad.numpy().copy()
        for n in self.fcw:
            g[n + "/weights"] = self.fcw[n].grad.numpy().copy()
            g[n + "/biases"] = self.fcb[n].grad.numpy().copy()
        return float(loss.detach()), logits.detach().numpy(), g

    def train_step(self, x, y, lr0, weight_decay, lr_factor=0.5, mask=None):
        if self._opt is None:
            self._opt = torch.optim.SGD(self.params_list(), lr=lr0, momentum=MOMENTUM)
        for grp in self._opt.param_groups:
            grp["lr"] = learning_rate(lr0, self.global_step, lr_factor)
        self._opt.zero_grad(set_to_none=True)
        logits = self.forward(x, True)
        loss = self.loss(logits, y, weight_decay, mask)
        loss.backward()
        self._opt.step()
        self.global_step += 1
        return float(loss.detach()), logits.detach().argmax(dim=3).numpy()
