"""Train steps behind the reference's outer loop (train_and_test.py), built from the drop-in modules.

* ``ClassifierTrainer`` — S1: FE → CLF → CE → backward → RMSprop×2 (train_and_test.py:148-171, no CPC term).
* ``JointTrainer``      — S2: one batch of the joint phase incl. GradNorm (train_and_test.py:539-766).

Differences from the reference are mechanical, not numerical:
  - the double ``loss_total.backward()`` (quirk Q3) is reproduced by ONE backward of
    Σ wᵢ·Lᵢ + 2·(a·cdan + b·fd_s + c·sl_t + d·sl_s) — same accumulated gradients, half the work;
  - GradNorm's numpy round-trip (:694-711) stays on the device (same fp32 formulas, no host sync);
  - per-batch ``.cpu()`` prints and feature dumps (:564-644) are not part of the step;
  - during GradNorm's five partial backward passes only the shared OS_block needs weight gradients, so
    every other conv skips its weight-gradient kernels (``ops.partial_backward``).
Data parallelism: ``dist.GradBucket`` all-reduces one flat fp32 gradient bucket over RCCL.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .cdan import CDAN, RandomLayer
from .cpc import CPC
from . import dist as _dist
from .dist import GradBucket
from .optim import FusedRMSprop, SharedStepAdam, rmsprop_step_many
from .os_cnn import OS_CNN, OS_CNN_res, build_layer_with_layer_parameter
from .structure import generate_layer_parameter_list, layer_parameter_list_input_change, out_channels
from .waveglow import WaveGlow, WaveGlowLoss
from .widgets import (AdversarialNetworkforCDAN, DimensionUnification, FeatureDiscriminatorforSource, NoiseTransfer,
                      ProbTransfer, wgan_loss)

MAX_KERNEL_SIZE = 89                                                          # train_and_test.py:40


def specs_for(length: int, in_channel: int):
    """(feature-extractor spec, classifier spec) exactly as train_and_test.py:38-53 derives them."""
    budgets = [8 * 128 * in_channel, 5 * 128 * 256 + 2 * 256 * 128]
    rf = min(int(length / 4), MAX_KERNEL_SIZE)
    fe = generate_layer_parameter_list(1, rf, budgets, in_channel)
    return fe, layer_parameter_list_input_change(fe, out_channels(fe[-1]))


def loss_coefficients(epoch: int) -> Tuple[float, float, float, float]:
    """(cdan, fd_s, sl_t, sl_s) coefficients by epoch (train_and_test.py:665-672)."""
    if epoch < 12:
        return 3, 3, 2, 2
    if epoch < 24:
        return 2, 3, 1.8, 1.5
    if epoch < 50:
        return 1.5, 2, 1.8, 1.8
    return 1.5, 1.5, 2.5, 2.5


# hipGraph capture must not be invalidated by other threads' HIP calls: with an RCCL communicator alive, PyTorch's
# watchdog thread polls events while the step is being captured ("global" mode would then abort the capture).
_CAPTURE_MODE = "thread_local"


class ClassifierTrainer:
    def __init__(self, length: int, in_channel: int, n_class: int, device, bucket: Optional[GradBucket] = None,
                 sync: str = "ddp"):
        self.sync = sync
        fe_spec, clf_spec = specs_for(length, in_channel)
        self.fe = OS_CNN_res(fe_spec).to(device)
        self.clf = OS_CNN(clf_spec, n_class).to(device)
        # capturable: the optimisers' step counters live on the device, so the step can be captured into a hipGraph
        self.opt_fe = FusedRMSprop(self.fe.parameters(), lr=0.001)
        self.opt_clf = FusedRMSprop(self.clf.parameters(), lr=0.003)
        self.bucket = bucket
        self.fe.train(); self.clf.train()
        self._graph = None

    def parameters(self) -> List[nn.Parameter]:
        return list(self.fe.parameters()) + list(self.clf.parameters())

    def step(self, x: torch.Tensor, y: torch.Tensor):
        with _dist.global_batch(self.bucket if self.sync == "global" else None):
            with ops.pack_cache():
                logits, _ = self.clf(self.fe(x))
                loss = F.cross_entropy(logits, y)
                loss.backward()
        if self.bucket is not None:
            self.bucket.all_reduce(self.parameters())
        rmsprop_step_many([self.opt_fe, self.opt_clf])
        self.opt_fe.zero_grad(set_to_none=True); self.opt_clf.zero_grad(set_to_none=True)
        return loss.detach(), logits.detach()

    # ---- single GPU: the whole step as one hipGraph (the eager step is launch-bound: ~300 launches for ~2 ms of GPU work)
    def capture(self, x: torch.Tensor, y: torch.Tensor, warmup: int = 3):
        if self.bucket is not None:
            raise RuntimeError("ClassifierTrainer.capture(): single-GPU only (the DP step has an eager all-reduce)")
        self._g_x, self._g_y = x.clone(), y.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(self._g_x, self._g_y)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, capture_error_mode=_CAPTURE_MODE):
            self._g_out = self.step(self._g_x, self._g_y)

    def replay(self, x: torch.Tensor, y: torch.Tensor):
        """One captured step on a new batch of the captured shape; returns (loss, logits) in static buffers."""
        if self._graph is None:
            raise RuntimeError("call capture() first")
        self._g_x.copy_(x); self._g_y.copy_(y)
        self._graph.replay()
        return self._g_out


@dataclass
class JointConfig:
    L_t: int = 512
    C_in_t: int = 1
    L_s: int = 512
    C_in_s: int = 1
    n_class_t: int = 4
    n_class_s: int = 4
    nf_flows: int = 3                                                         # WaveGlow(3, C, 120) :71
    nf_channels: int = 120
    cpc_hidden: int = 64                                                      # CPC(C, 64, L//2) :131
    cdan_dim: int = 1024                                                      # :75-77
    ad_hidden: int = 1024
    dropout_p: float = 0.2
    nf_end_std: float = 0.0       # >0: draw WN.end (zero-init in the reference) so the flow is non-degenerate


class JointTrainer:
    MODULES = ("fe_t", "clf_t", "fe_s", "dimunif", "clf_s", "probtransfer", "nf", "noise", "ad_net", "fd_s", "cpc")
    LRS = {"fe_t": 0.001, "clf_t": 0.003, "fe_s": 0.001, "dimunif": 0.001, "clf_s": 0.003, "probtransfer": 0.001,
           "nf": 0.001, "noise": 0.005, "ad_net": 0.001, "fd_s": 0.001}       # train_and_test.py:97-106

    def __init__(self, cfg: JointConfig, device, bucket: Optional[GradBucket] = None, fe_t_spec=None, clf_spec=None,
                 fe_s_spec=None, sync: str = "ddp"):
        """``sync`` (with a bucket): "ddp" = per-rank batch statistics (SURVEY §8e mode A); "global" = every
        batch-coupled quantity over the samples of all ranks (mode B, eager only) — N ranks reproduce the
        single-process step on the concatenated batch."""
        if sync not in ("ddp", "global"):
            raise ValueError(f"sync must be 'ddp' or 'global', got {sync!r}")
        self.cfg, self.device, self.bucket, self.sync = cfg, device, bucket, sync
        if fe_t_spec is None:
            fe_t_spec, clf_spec = specs_for(cfg.L_t, cfg.C_in_t)
            fe_s_spec, _ = specs_for(cfg.L_s, cfg.C_in_s)
        C, C_s = out_channels(fe_t_spec[-1]), out_channels(fe_s_spec[-1])
        m: Dict[str, nn.Module] = {}
        m["fe_t"] = OS_CNN_res(fe_t_spec)
        m["clf_t"] = OS_CNN(clf_spec, cfg.n_class_t)
        m["fe_s"] = OS_CNN_res(fe_s_spec)
        m["dimunif"] = DimensionUnification(C_s, C, cfg.L_s, cfg.L_t)
        m["clf_s"] = OS_CNN(clf_spec, cfg.n_class_s)
        m["probtransfer"] = ProbTransfer(m["clf_s"].length_before_classification)
        m["nf"] = WaveGlow(cfg.nf_flows, C, cfg.nf_channels)
        if cfg.nf_end_std > 0:
            for wn in m["nf"].WN:
                wn.end.weight.data.normal_(0, cfg.nf_end_std)
                wn.end.bias.data.normal_(0, cfg.nf_end_std)
        m["noise"] = NoiseTransfer(C, cfg.L_t)
        self.random_layer = RandomLayer([C * cfg.L_t, cfg.n_class_t], cfg.cdan_dim)
        m["ad_net"] = AdversarialNetworkforCDAN(cfg.cdan_dim, cfg.ad_hidden)
        m["ad_net"].dropout1.p = m["ad_net"].dropout2.p = cfg.dropout_p
        m["fd_s"] = FeatureDiscriminatorforSource(m["clf_s"].length_before_classification)
        m["cpc"] = CPC(C, cfg.cpc_hidden, cfg.L_t // 2)
        self.m = {k: v.to(device) for k, v in m.items()}
        self.random_layer = self.random_layer.to(device)
        self.nf_loss = WaveGlowLoss()
        # ten RMSprops (one learning rate per module) stepped by the same single-pass multi-tensor launches (optim.FusedRMSprop)
        self.opts = {k: FusedRMSprop(self.m[k].parameters(), lr=lr) for k, lr in self.LRS.items()}
        # CPC = 516 small tensors: torch's capturable Adam spends ~4 k tiny kernels per step on per-parameter step
        # counters; same update with one shared device counter (optim.SharedStepAdam)
        self.opt_cpc = SharedStepAdam(self.m["cpc"].parameters(), lr=0.002)
        self.w_t = nn.Parameter(torch.tensor([2.0, 5.0], device=device))         # :501-505
        self.w_s = nn.Parameter(torch.tensor([2.0, 2.0, 4.0], device=device))
        self.opt_w_t = torch.optim.Adam([self.w_t], lr=0.0002, capturable=True)
        self.opt_w_s = torch.optim.Adam([self.w_s], lr=0.001, capturable=True)
        self.init_t = self.init_s = None
        self.alpha = 3
        self.on_grads_ready = None                                            # test hook: called before the optimisers step
        self._side = torch.cuda.Stream(device=device)                         # launch-bound side chains (CPC)
        for mod in self.m.values():
            mod.train()
        # GradNorm differentiates the shared OS_blocks only: their convs keep weight gradients in partial passes
        for key in ("fe_t", "fe_s"):
            for layer in self.m[key].return_last_layer().layer_list:
                layer.spec.always_weight_grad = True

    # ------------------------------------------------------------------ helpers
    def parameters(self) -> List[nn.Parameter]:
        return [p for k in self.MODULES for p in self.m[k].parameters()]

    def load_params(self, mods: Dict[str, Dict[str, torch.Tensor]], mats=None) -> None:
        """Load per-module state (reference state_dict key names) — used by the parity tests."""
        for k, sd in mods.items():
            self.m[k].load_state_dict({n: v.detach().to(self.device) for n, v in sd.items()})
        if mats is not None:
            self.random_layer.random_matrix = [t.to(self.device) for t in mats]
            self.random_layer._transposed = {}

    # ------------------------------------------------------------------ state snapshot (in-place restore keeps addresses)
    def _state_tensors(self):
        out = {}
        for k in self.MODULES:
            for n, t in self.m[k].state_dict().items():
                out[f"m.{k}.{n}"] = t
        opts = dict(self.opts)
        opts.update({"cpc": self.opt_cpc, "w_t": self.opt_w_t, "w_s": self.opt_w_s})
        for k, o in opts.items():
            for gi, group in enumerate(o.param_groups):
                for pi, p in enumerate(group["params"]):
                    for n, t in o.state.get(p, {}).items():
                        if isinstance(t, torch.Tensor):
                            out[f"o.{k}.{gi}.{pi}.{n}"] = t
        out["w_t"], out["w_s"] = self.w_t.data, self.w_s.data
        out["noise.target_avg"], out["noise.source_avg"] = self.m["noise"].target_avg, self.m["noise"].source_avg
        if self.init_t is not None:
            out["init_t"], out["init_s"] = self.init_t, self.init_s
        return out

    def snapshot(self):
        """Copy of every tensor the step mutates (parameters, BN buffers, optimiser moments, GradNorm weights,
        NoiseTransfer sums) plus the host-side counters."""
        host = {"noise": (self.m["noise"].time, self.m["noise"].cal_num_target, self.m["noise"].cal_num_source),
                "ad": self.m["ad_net"].iter_num, "fd": self.m["fd_s"].iter_num}
        return {"t": {k: v.detach().clone() for k, v in self._state_tensors().items()}, "host": host}

    def restore(self, snap) -> None:
        cur = self._state_tensors()
        with torch.no_grad():
            for k, v in snap["t"].items():
                cur[k].copy_(v)
        n = self.m["noise"]
        n.time, n.cal_num_target, n.cal_num_source = snap["host"]["noise"]
        self.m["ad_net"].iter_num, self.m["fd_s"].iter_num = snap["host"]["ad"], snap["host"]["fd"]

    # ------------------------------------------------------------------ trainer-state checkpoint (resume == uninterrupted)
    def state_dict(self) -> dict:
        """Everything one more step depends on, on the CPU: the eleven modules, all thirteen optimisers, the GradNorm
        weights and their reference losses, NoiseTransfer's running sums and counters (Q5), the GRL call counters (Q7),
        the fixed CDAN random matrices and WaveGlow's cached — possibly stale — inverses (Q2).  The reference keeps
        none of this across runs (utils.py:9-25 saves the classification modules only); a captured-graph trainer that
        cannot resume would be a gap of this build, not of the reference."""
        cpu = lambda t: t.detach().cpu().clone()
        cpc = self.opt_cpc
        noise = self.m["noise"]

        def opt_cpu(o):
            # Optimizer.state_dict() aliases the live (device) moment tensors: copy them out, or an in-memory snapshot
            # followed by more steps would restore advanced moments
            sd = o.state_dict()
            return {"state": {i: {n: (cpu(v) if isinstance(v, torch.Tensor) else v) for n, v in st.items()}
                              for i, st in sd["state"].items()},
                    "param_groups": [dict(g) for g in sd["param_groups"]]}
        return {
            "modules": {k: {n: cpu(v) for n, v in self.m[k].state_dict().items()} for k in self.MODULES},
            "opts": {k: opt_cpu(o) for k, o in self.opts.items()},
            "opt_w_t": opt_cpu(self.opt_w_t), "opt_w_s": opt_cpu(self.opt_w_s),
            "opt_cpc": {"step": [cpu(g["step"]) for g in cpc.param_groups],
                        "exp_avg": [[cpu(cpc.state[p]["exp_avg"]) for p in g["params"]] for g in cpc.param_groups],
                        "exp_avg_sq": [[cpu(cpc.state[p]["exp_avg_sq"]) for p in g["params"]] for g in cpc.param_groups]},
            "w_t": cpu(self.w_t), "w_s": cpu(self.w_s),
            "init_t": None if self.init_t is None else cpu(self.init_t),
            "init_s": None if self.init_s is None else cpu(self.init_s),
            "noise": {"target_avg": cpu(noise.target_avg), "source_avg": cpu(noise.source_avg), "time": noise.time,
                      "cal_num_target": noise.cal_num_target, "cal_num_source": noise.cal_num_source},
            "grl": {"ad_net": self.m["ad_net"].iter_num, "fd_s": self.m["fd_s"].iter_num},
            "random_matrix": [cpu(t) for t in self.random_layer.random_matrix],
            "w_inverse": [cpu(c.W_inverse) if hasattr(c, "W_inverse") else None for c in self.m["nf"].convinv],
        }

    def load_state_dict(self, sd: dict) -> None:
        """Inverse of ``state_dict``.  A captured graph holds the old buffers' addresses only for parameters and
        optimiser moments that are restored IN PLACE here, but re-capture after loading anyway (GRL coefficients and
        the epoch's loss coefficients are baked into a capture)."""
        dev = self.device
        for k in self.MODULES:
            self.m[k].load_state_dict({n: v.to(dev) for n, v in sd["modules"][k].items()}, strict=True)
        for k, o in self.opts.items():
            o.load_state_dict(sd["opts"][k])
        self.opt_w_t.load_state_dict(sd["opt_w_t"]); self.opt_w_s.load_state_dict(sd["opt_w_s"])
        cpc = self.opt_cpc
        with torch.no_grad():
            for gi, g in enumerate(cpc.param_groups):
                g["step"].copy_(sd["opt_cpc"]["step"][gi])
                for pi, p in enumerate(g["params"]):
                    cpc.state[p]["exp_avg"].copy_(sd["opt_cpc"]["exp_avg"][gi][pi])
                    cpc.state[p]["exp_avg_sq"].copy_(sd["opt_cpc"]["exp_avg_sq"][gi][pi])
            self.w_t.copy_(sd["w_t"]); self.w_s.copy_(sd["w_s"])
            noise = self.m["noise"]
            noise.target_avg.copy_(sd["noise"]["target_avg"]); noise.source_avg.copy_(sd["noise"]["source_avg"])
        noise.time, noise.cal_num_target, noise.cal_num_source = (sd["noise"][k] for k in ("time", "cal_num_target", "cal_num_source"))
        self.init_t = None if sd["init_t"] is None else sd["init_t"].to(dev)
        self.init_s = None if sd["init_s"] is None else sd["init_s"].to(dev)
        self.m["ad_net"].iter_num, self.m["fd_s"].iter_num = sd["grl"]["ad_net"], sd["grl"]["fd_s"]
        self.random_layer.random_matrix = [t.to(dev) for t in sd["random_matrix"]]
        self.random_layer._transposed = {}
        for c, w in zip(self.m["nf"].convinv, sd["w_inverse"]):
            if w is not None:
                c.W_inverse = w.to(dev)
            elif hasattr(c, "W_inverse"):
                del c.W_inverse

    def save_state(self, path: str) -> None:
        torch.save(self.state_dict(), path)

    def load_state(self, path: str) -> None:
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=False))

    # ------------------------------------------------------------------ forward (train_and_test.py:547-603)
    def forward_losses(self, x_t, y_t, x_s, y_s, t_samples=(None, None), noise_ratios=None):
        m = self.m
        # log|det W| of the flows' 1x1 weights: three single-workgroup launches that depend on the weights only — on the side stream,
        # beside the feature extractors, instead of in the chain of the first flow pass
        self._side.wait_stream(torch.cuda.current_stream())
        m["nf"].prefetch_logdets(self._side)
        feat_t = m["fe_t"](x_t)
        feat_s = m["dimunif"](m["fe_s"](x_s))
        # The two CPC losses are ~3 k tiny launches (MIOpen's GRU runs step by step) that would leave the chip idle in
        # the captured graph's single chain.  Fork them onto a side stream: they (and their backward, which autograd
        # runs on the stream of the forward op) overlap the WaveGlow passes; joined before the losses are summed.
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            sl_t = m["cpc"](feat_t, t_samples[0])
            sl_s = m["cpc"](feat_s, t_samples[1])
        out_t, out_s = m["nf"](feat_t), m["nf"](feat_s)
        nf_t, nf_s = self.nf_loss(out_t), self.nf_loss(out_s)
        z_s2t = m["noise"](out_t[0], out_s[0], noise_ratios)
        feat_s2t = m["nf"].infer(z_s2t)
        logit_t, pool_t = m["clf_t"](feat_t)
        m["clf_t"].eval()                                                     # :584-586
        logit_s2t, pool_s2t = m["clf_t"](feat_s2t)
        m["clf_t"].train()
        logit_s, pool_s = m["clf_s"](feat_s)
        ce_t, ce_s = F.cross_entropy(logit_t, y_t), F.cross_entropy(logit_s, y_s)
        cdan = CDAN(feat_t, feat_s2t, logit_t, logit_s2t, m["ad_net"], self.random_layer)
        tr_t, tr_s2t = m["probtransfer"](pool_t), m["probtransfer"](pool_s2t)
        ce_s2t2s = F.cross_entropy(ops.linear_act(tr_s2t, m["clf_s"].hidden), y_s)
        fd = wgan_loss(m["fd_s"](tr_t), m["fd_s"](tr_s2t), m["fd_s"](pool_s))
        main.wait_stream(self._side)
        for t in (feat_t, feat_s):                                             # consumed on the side stream too
            t.record_stream(self._side)
        losses = {"nf_t": nf_t, "nf_s": nf_s, "ce_t": ce_t, "sl_t": sl_t, "ce_s": ce_s, "sl_s": sl_s, "cdan": cdan,
                  "ce_s2t2s": ce_s2t2s, "fd_s": fd}
        aux = {"logit_t": logit_t, "logit_s": logit_s, "logit_s2t": logit_s2t, "feat_t": feat_t, "feat_s2t": feat_s2t}
        return losses, aux

    # ------------------------------------------------------------------ pre-training phases (train_and_test.py:141-494)
    PHASES = {                                                                       # phase -> optimisers stepped
        "target_pretrain": ("fe_t", "clf_t", "cpc"),                                 # :143-171  CE_t + CPC_t
        "source_pretrain": ("fe_s", "dimunif", "clf_s"),                             # :183-209  CE_s
        "ssl_with_ce": ("fe_t", "clf_t", "cpc", "fe_s", "dimunif", "clf_s"),         # :232-275  every 50th epoch
        "ssl": ("fe_t", "cpc", "fe_s", "dimunif"),                                   # :296-348
        "nf_with_ce": ("fe_t", "clf_t", "fe_s", "dimunif", "clf_s", "nf", "cpc"),    # :388-431  every 75th epoch
        "nf": ("fe_t", "fe_s", "dimunif", "nf"),                                     # :457-494  features detached
    }

    def phase_losses(self, phase: str, x_t, y_t, x_s, y_s, t_samples=(None, None)):
        """(total, losses) of one batch of a pre-training phase — sub-graphs of the joint step on the same modules.
        "ssl" runs both classifiers in train mode (their BatchNorm running statistics move) although nothing of theirs
        is in the total; "nf" detaches the features, so only the flow receives gradients."""
        if phase not in self.PHASES:
            raise ValueError(f"unknown phase {phase!r}; one of {sorted(self.PHASES)}")
        m, L = self.m, {}
        if phase == "target_pretrain":
            feat_t = m["fe_t"](x_t)
            L["sl_t"] = m["cpc"](feat_t, t_samples[0])
            L["ce_t"] = F.cross_entropy(m["clf_t"](feat_t)[0], y_t)
            return L["ce_t"] + L["sl_t"], L
        if phase == "source_pretrain":
            feat_s = m["dimunif"](m["fe_s"](x_s))
            L["ce_s"] = F.cross_entropy(m["clf_s"](feat_s)[0], y_s)
            return L["ce_s"], L
        feat_t = m["fe_t"](x_t)
        feat_s = m["dimunif"](m["fe_s"](x_s))
        if phase == "nf":
            feat_t, feat_s = feat_t.detach(), feat_s.detach()
        else:
            L["sl_t"] = m["cpc"](feat_t, t_samples[0])
            L["ce_t"] = F.cross_entropy(m["clf_t"](feat_t)[0], y_t)
            L["sl_s"] = m["cpc"](feat_s, t_samples[1])
            L["ce_s"] = F.cross_entropy(m["clf_s"](feat_s)[0], y_s)
        if phase == "ssl_with_ce":
            return L["sl_t"] + L["sl_s"] + 0.8 * L["ce_t"] + 1.2 * L["ce_s"], L
        if phase == "ssl":
            return L["sl_t"] + L["sl_s"], L
        L["nf_t"], L["nf_s"] = self.nf_loss(m["nf"](feat_t)), self.nf_loss(m["nf"](feat_s))
        if phase == "nf_with_ce":
            return L["nf_t"] + L["nf_s"] + 5 * L["ce_t"] + 5 * L["ce_s"] + 3 * L["sl_t"] + 3 * L["sl_s"], L
        return L["nf_t"] + L["nf_s"], L

    def phase_step(self, phase: str, x_t, y_t, x_s, y_s, t_samples=(None, None)):
        """One batch of a pre-training phase: forward, backward, the phase's optimisers, zero_grad (eager)."""
        with _dist.global_batch(self.bucket if self.sync == "global" else None):
            with ops.pack_cache(), self.m["nf"].shared_fold(), self.m["cpc"].shared_stack():
                total, L = self.phase_losses(phase, x_t, y_t, x_s, y_s, t_samples)
                for o in self.opts.values():
                    o.zero_grad(set_to_none=True)
                self.opt_cpc.zero_grad(set_to_none=True)
                total.backward()
        if self.bucket is not None:
            self.bucket.all_reduce(self.parameters())
        if self.on_grads_ready is not None:
            self.on_grads_ready()
        rmsprop_step_many([self.opts[k] for k in self.PHASES[phase] if k != "cpc"])
        if "cpc" in self.PHASES[phase]:
            self.opt_cpc.step()
        report = {k: v.detach() for k, v in L.items()}
        report["total"] = total.detach()
        return report

    # ------------------------------------------------------------------ one optimisation step (:645-766)
    def step(self, x_t, y_t, x_s, y_s, epoch: int = 0, t_samples=(None, None)):
        """Eager step.  ``t_samples``: the two CPC start indices (drawn like the reference if None)."""
        ratios = self.m["noise"].advance(x_t.size(0), x_s.size(0))
        return self._step_body(x_t, y_t, x_s, y_s, epoch, t_samples, ratios)

    def _step_body(self, x_t, y_t, x_s, y_s, epoch, t_samples, noise_ratios):
        """Everything device-side and shape-static, so it runs eagerly or under hipGraph capture unchanged.
        Three parts with the step's only collectives between them (so a captured step never contains RCCL):
        A1 = forward + the full backward;  [the gradient bucket's all-reduce starts on a side stream]
        A2 = GradNorm's partial backward passes (they never touch ``.grad``: they overlap the all-reduce);
        [wait for the bucket; average the 10 GradNorm scalars]  B = GradNorm weight update + optimisers."""
        with _dist.global_batch(self.bucket if self.sync == "global" else None), self._step_scope():
            state = self._step_part_a1(x_t, y_t, x_s, y_s, epoch, t_samples, noise_ratios)
            self._reduce_begin()
            mid = self._step_part_a2(state)
        self._reduce_end(mid)
        return self._step_part_b(mid)

    def _step_scope(self):
        """One scope for forward and every backward pass of a step: weights packed once, the flow's weight-norm fold and the
        CPC predictor stack built once."""
        import contextlib
        st = contextlib.ExitStack()
        st.enter_context(ops.pack_cache())
        st.enter_context(self.m["nf"].shared_fold())
        st.enter_context(self.m["cpc"].shared_stack())
        return st

    def _step_part_a1(self, x_t, y_t, x_s, y_s, epoch, t_samples, noise_ratios):
        if True:
            L, aux = self.forward_losses(x_t, y_t, x_s, y_s, t_samples, noise_ratios)
            lt = torch.stack([L["nf_t"], L["ce_t"]])
            ls = torch.stack([L["nf_s"], L["ce_s"], L["ce_s2t2s"]])
            a, b, c, d = loss_coefficients(epoch)
            # Q3: first backward + second backward (weights zeroed) == Σ wᵢ∇Lᵢ + 2·(a∇cdan + b∇fd + c∇sl_t + d∇sl_s)
            total = torch.sum(self.w_t.detach() * lt) + torch.sum(self.w_s.detach() * ls) \
                + 2.0 * (a * L["cdan"] + b * L["fd_s"] + c * L["sl_t"] + d * L["sl_s"])
            for o in self.opts.values():
                o.zero_grad(set_to_none=True)
            self.opt_cpc.zero_grad(set_to_none=True)
            total.backward(retain_graph=True)
        return {"L": L, "aux": aux, "lt": lt, "ls": ls}

    def _step_part_a2(self, state):
        L, aux, lt, ls = state["L"], state["aux"], state["lt"], state["ls"]
        if True:
            # GradNorm (:682-690): per-loss gradient norms over the 12 shared tensors
            sh_t = list(self.m["fe_t"].return_last_layer().parameters())
            sh_s = list(self.m["fe_s"].return_last_layer().parameters())
            # Differentiate the loss tensors themselves, not lt[i] / ls[i]: a select of the stacked vector sends a ZERO
            # cotangent down every other loss's graph (autograd does not prune zeros), i.e. 8 WaveGlow backward
            # traversals per step where 4 carry anything (ce_t / ce_s never touch the flow).  Same values, half the work.
            with ops.partial_backward():
                g_t = [torch.autograd.grad(L[k], sh_t, retain_graph=True) for k in ("nf_t", "ce_t")]
                g_s = [torch.autograd.grad(L[k], sh_s, retain_graph=(k != "ce_s2t2s")) for k in ("nf_s", "ce_s", "ce_s2t2s")]
            if _dist.global_batch_active():
                # mode B: the norms are those of the GLOBAL per-loss gradients (mean over ranks), not means of norms
                flat = torch.cat([g.reshape(-1) for gs in g_t + g_s for g in gs])
                flat = flat / _dist.sum_over_ranks_(flat)
                it = iter(torch.split(flat, [g.numel() for gs in g_t + g_s for g in gs]))
                g_t = [[next(it).view_as(g) for g in gs] for gs in g_t]
                g_s = [[next(it).view_as(g) for g in gs] for gs in g_s]
            # Σ_θ ‖∂L_i/∂θ‖₂ per loss (:685-690): all 5 × 12 norms as ONE multi-tensor launch (60 separate reductions before)
            norms = torch.stack(torch._foreach_norm([g for gs in g_t + g_s for g in gs])).view(len(g_t) + len(g_s), -1).sum(dim=1)
            base_t, base_s = norms[: len(g_t)], norms[len(g_t):]
        report = {k: v.detach() for k, v in L.items()}
        report.update({k: v.detach() for k, v in aux.items()})
        # scalars that must be identical on every rank: loss values and gradient-norm bases (10 floats)
        scal = torch.cat([lt.detach(), ls.detach(), base_t, base_s]).contiguous()
        return {"report": report, "scal": scal}

    def _step_part_a(self, x_t, y_t, x_s, y_s, epoch, t_samples, noise_ratios):
        with self._step_scope():
            return self._step_part_a2(self._step_part_a1(x_t, y_t, x_s, y_s, epoch, t_samples, noise_ratios))

    def _reduce_begin(self) -> None:
        if self.bucket is not None:
            self.bucket.all_reduce_begin(self.parameters())

    def _reduce_end(self, mid) -> None:
        if self.bucket is None:
            return
        self.bucket.all_reduce_end()
        mid["scal"].copy_(self.bucket.mean_scalars(mid["scal"]))

    def _step_part_b(self, mid):
        scal = mid["scal"]
        lt_v, ls_v, base_t, base_s = scal[0:2], scal[2:5], scal[5:7], scal[7:10]
        if self.init_t is None:                                               # :658-664
            self.init_t, self.init_s = torch.sigmoid(lt_v).clone(), torch.sigmoid(ls_v).clone()
        # ‖wᵢ·g‖ = |wᵢ|·‖g‖, so the norms are differentiable functions of w alone (:685-715)
        nt, ns = torch.abs(self.w_t) * base_t, torch.abs(self.w_s) * base_s
        ratio_t, ratio_s = torch.sigmoid(lt_v) / self.init_t, torch.sigmoid(ls_v) / self.init_s
        inv_t, inv_s = ratio_t / ratio_t.mean(), ratio_s / ratio_s.mean()
        const_t = (nt.detach().mean() * inv_t ** self.alpha).detach()
        const_s = (ns.detach().mean() * inv_s ** self.alpha).detach()
        g_w_t = torch.autograd.grad(torch.sum(torch.abs(nt - const_t)), self.w_t)[0]
        g_w_s = torch.autograd.grad(torch.sum(torch.abs(ns - const_s)), self.w_s)[0]
        for w, g in ((self.w_t, g_w_t), (self.w_s, g_w_s)):                   # static .grad buffers (graph-safe)
            if w.grad is None:
                w.grad = torch.zeros_like(w)
            w.grad.copy_(g)
        if self.on_grads_ready is not None:
            self.on_grads_ready()
        self.opt_w_t.step(); self.opt_w_s.step()
        rmsprop_step_many(list(self.opts.values()))
        self.opt_cpc.step()
        with torch.no_grad():                                                 # :756-766
            self.w_t.clamp_(min=0.0)
            self.w_t.mul_(7 / torch.sum(self.w_t))
            self.w_s.clamp_(min=0.0)
            self.w_s.mul_(8 / torch.sum(self.w_s))
            for p in self.m["ad_net"].parameters():
                p.clamp_(-0.0005, 0.0005)
            for p in self.m["fd_s"].parameters():
                p.clamp_(-0.01, 0.01)
        report = dict(mid["report"])
        report.update({"w_t": self.w_t.detach().clone(), "w_s": self.w_s.detach().clone(),
                       "norms_t": nt.detach(), "norms_s": ns.detach()})
        return report

    # ------------------------------------------------------------------ hipGraph: capture once, replay per step
    def capture(self, x_t, y_t, x_s, y_s, epoch: int = 0, warmup: int = 11):
        """Capture one whole step (≈12 k launches: forward, GradNorm partial backwards, backward, optimisers) into
        hipGraphs.  Per-step inputs live in static device buffers that ``replay`` refreshes: the batch, the two CPC
        start indices and NoiseTransfer's two accumulation ratios.  The GRL coefficients are Python floats baked in
        at capture, so the warm-up runs until their call counters saturate (20 calls = 10 steps — quirk Q7); the
        epoch-dependent loss coefficients are baked too: re-capture when ``loss_coefficients(epoch)`` changes.
        Single GPU: one graph.  Data parallel: three graphs (A1 forward + backward, A2 GradNorm's partial passes, B update)
        with the eager RCCL collectives between them, the gradient all-reduce overlapping A2 on a side stream."""
        if self.sync == "global" and self.bucket is not None and self.bucket.world > 1:
            raise RuntimeError("sync='global' (mode B) puts collectives inside autograd: run it eagerly with step()")
        dev = self.device
        self._g_in = {"x_t": x_t.clone(), "y_t": y_t.clone(), "x_s": x_s.clone(), "y_s": y_s.clone(),
                      "t": torch.zeros(2, dtype=torch.int32, device=dev), "r": torch.ones(2, device=dev)}
        self._g_epoch = epoch
        T_half = max(1, (self.cfg.L_t // 2) // 2)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                         # eager warm-up on a side stream
            for _ in range(warmup):
                self._replay_inputs(x_t, y_t, x_s, y_s, (int(torch.randint(T_half, (1,))), int(torch.randint(T_half, (1,)))))
                self._graph_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.m["ad_net"].iter_num < self.m["ad_net"].max_iter or self.m["fd_s"].iter_num < self.m["fd_s"].max_iter:
            raise RuntimeError("capture(): GRL call counters not saturated yet; increase warmup")
        self._replay_inputs(x_t, y_t, x_s, y_s, (0, 0))
        if self.bucket is None:
            self._graphs = [torch.cuda.CUDAGraph()]
            with torch.cuda.graph(self._graphs[0], capture_error_mode=_CAPTURE_MODE):
                self._g_out = self._graph_body()
        else:
            ga1, ga2, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            gi = self._g_in
            with self._step_scope():                                          # packed weights of A1 are reused by A2
                with torch.cuda.graph(ga1, capture_error_mode=_CAPTURE_MODE):
                    state = self._step_part_a1(gi["x_t"], gi["y_t"], gi["x_s"], gi["y_s"], self._g_epoch, (gi["t"][0], gi["t"][1]),
                                               (gi["r"][0], gi["r"][1]))
                pool = ga1.pool()
                self._reduce_begin()                                          # eager, on the bucket's side stream
                with torch.cuda.graph(ga2, pool=pool, capture_error_mode=_CAPTURE_MODE):
                    self._g_mid = self._step_part_a2(state)
                del state                                                     # the autograd graph of the captured step
            self._reduce_end(self._g_mid)                                     # eager; also fixes the bucket's buffer
            with torch.cuda.graph(gb, pool=pool, capture_error_mode=_CAPTURE_MODE):
                self._g_out = self._step_part_b(self._g_mid)
            self._graphs = [ga1, ga2, gb]
        return self

    def _graph_body(self):
        gi = self._g_in
        return self._step_body(gi["x_t"], gi["y_t"], gi["x_s"], gi["y_s"], self._g_epoch, (gi["t"][0], gi["t"][1]),
                               (gi["r"][0], gi["r"][1]))

    def _replay_inputs(self, x_t, y_t, x_s, y_s, t_samples):
        gi = self._g_in
        for k, v in (("x_t", x_t), ("y_t", y_t), ("x_s", x_s), ("y_s", y_s)):
            if v is not gi[k]:
                gi[k].copy_(v, non_blocking=True)
        ratios = self.m["noise"].advance(x_t.size(0), x_s.size(0))
        host = torch.tensor([float(t_samples[0]), float(t_samples[1]), ratios[0], ratios[1]], dtype=torch.float64)
        dev = host.to(self.device, non_blocking=True)
        gi["t"].copy_(dev[:2])
        gi["r"].copy_(dev[2:])

    def replay(self, x_t, y_t, x_s, y_s, t_samples):
        """One step through the captured graph(s); returns the (static) report tensors."""
        self._replay_inputs(x_t, y_t, x_s, y_s, t_samples)
        self._graphs[0].replay()
        if len(self._graphs) == 3:
            self._reduce_begin()                                              # the bucket's all-reduce, on its side stream ...
            self._graphs[1].replay()                                          # ... under GradNorm's partial backward passes
            self._reduce_end(self._g_mid)
            self._graphs[2].replay()
        return self._g_out
