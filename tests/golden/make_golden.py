#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the fixtures it writes are committed and
travel to the GPU box, this script's dependency on the reference does not.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Procedure: `speech_decoding/models.py` imports three packages that are absent here
and do no arithmetic on the path (`termcolor` for coloured prints, `mne`/`mne_bids` for the sensor
lookup).  Empty stand-in modules are registered in `sys.modules` of THIS process only, and the
reference's `ch_locations_2d` is replaced by a seeded synthetic position table normalised exactly as
layout.py:38-41.  Everything else — every number in the fixtures — is computed by the reference code.

Fixtures:
  e2e_small.npz      reduced dims, train mode: full state, inputs, per-stage activations, Z, logits,
                     loss, every gradient (incl. complex z.grad and temp.grad), BN running stats
                     after one step, all parameters after two Adam steps, eval-mode Z/loss.
  classifier.npz     reference `Classifier` (the B² Python loop) on a fixed (Z, Y), B = 24.
  spot_208.npz       full dims (C=208, S=27, B=8, T=360): seeded state (oracle.seeded_params, NOT
  spot_60.npz        stored), sampled Z / gradient entries, logits, loss.  Ditto C=60, S=1.
  collate.npz        reference preproc_utils.baseline_correction_single + scaleAndClamp on a fixed batch.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SD_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

for name in ("termcolor", "mne", "mne_bids", "omegaconf"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        if name == "termcolor":
            m.cprint = lambda *a, **k: None
        if name == "omegaconf":             # only type names / a context manager, used by check_preprocs (not called)
            m.DictConfig = dict
            m.open_dict = lambda cfg: cfg
        sys.modules[name] = m

import speech_decoding.models as ref_models            # noqa: E402  (the reference)
from speech_decoding.utils.loss import CLIPLoss as RefCLIPLoss  # noqa: E402

from oracle import brain_oracle as O                    # noqa: E402  (input generators only)


class Args(dict):
    """attribute + item access, like the Hydra DictConfig the reference receives."""
    __getattr__ = dict.__getitem__


def make_args(C, S, D1, D2, F, K, last4layers):
    return Args(num_subjects=S, D1=D1, D2=D2, F=F, K=K, dataset="Gwilliams2022", d_drop=0.1,
                root_dir="/nonexistent", preprocs={"last4layers": last4layers}, reduction="mean",
                init_temperature=5.1, num_channels=C)


def build_reference(args, loc):
    ref_models.ch_locations_2d = lambda a: loc.clone()
    enc = ref_models.BrainEncoder(args)
    loss = RefCLIPLoss(args)
    return enc, loss


def peek_drop_centre(seed, C):
    np.random.seed(seed)
    c = int(np.random.randint(C))
    np.random.seed(seed)           # the reference's own draw at models.py:81 will repeat it
    return c


def state_to_np(sd, prefix):
    out = {}
    for k, v in sd.items():
        if v.is_complex():
            out[f"{prefix}{k}@re"] = v.real.detach().numpy().copy()
            out[f"{prefix}{k}@im"] = v.imag.detach().numpy().copy()
        else:
            out[f"{prefix}{k}"] = v.detach().numpy().copy()
    return out


def gen_e2e_small():
    C, S, D1, D2, F, K, T, B = 12, 3, 16, 24, 32, 4, 40, 6
    torch.manual_seed(0)
    loc = O.synthetic_positions(C, seed=7)
    args = make_args(C, S, D1, D2, F, K, last4layers=False)
    enc, lossf = build_reference(args, loc)
    # non-trivial BN affine so that gamma/beta paths are exercised
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for k in range(5):
            for j in (0, 1):
                bn = getattr(enc.conv_blocks, f"conv{k}")
                bn = getattr(bn, f"batchnorm{j}")
                bn.weight.copy_(torch.rand(D2, generator=g) + 0.5)
                bn.bias.copy_(torch.rand(D2, generator=g) - 0.5)
    X, Y, subj = O.synthetic_batch(B, C, T, F, S, seed=1234)
    out = {"dims": np.array([C, S, D1, D2, F, K, T, B]), "loc": loc.numpy(), "X": X.numpy(),
           "Y": Y.numpy(), "subject_idxs": subj.numpy()}
    out.update(state_to_np(enc.state_dict(), "init/"))
    out["init/temp"] = lossf.temp.detach().numpy().copy()

    # per-stage activations through forward hooks
    acts = {}
    hooks = [enc.subject_block.spatial_attention.register_forward_hook(
                 lambda m, i, o: acts.__setitem__("act/spatial_attention", o.detach().numpy().copy())),
             enc.subject_block.register_forward_hook(
                 lambda m, i, o: acts.__setitem__("act/subject_block", o.detach().numpy().copy())),
             enc.conv_final1.register_forward_hook(
                 lambda m, i, o: acts.__setitem__("act/conv_final1_pre_gelu", o.detach().numpy().copy()))]
    for k in range(5):
        hooks.append(getattr(enc.conv_blocks, f"conv{k}").register_forward_hook(
            lambda m, i, o, k=k: acts.__setitem__(f"act/conv_block{k}", o.detach().numpy().copy())))

    opt = torch.optim.Adam(list(enc.parameters()) + list(lossf.parameters()), lr=3e-4)
    enc.train()
    lossf.train()
    centres = []
    for step in range(2):
        centres.append(peek_drop_centre(100 + step, C))
        Z = enc(X, subj)
        logits, loss = lossf(Y, Z, return_logits=True)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            for h in hooks:
                h.remove()
            out.update(acts)
            out["step0/Z"] = Z.detach().numpy().copy()
            out["step0/logits"] = logits.detach().numpy().copy()
            out["step0/loss"] = loss.detach().numpy().copy()
            for n, p in enc.named_parameters():
                if p.grad.is_complex():
                    out[f"grad/{n}@re"] = p.grad.real.numpy().copy()
                    out[f"grad/{n}@im"] = p.grad.imag.numpy().copy()
                else:
                    out[f"grad/{n}"] = p.grad.numpy().copy()
            out["grad/temp"] = lossf.temp.grad.numpy().copy()
            out.update({k: v for k, v in state_to_np(enc.state_dict(), "after1fwd/").items()
                        if "running" in k or "num_batches" in k})
        else:
            out["step1/loss"] = loss.detach().numpy().copy()
        opt.step()
    out["drop_centres"] = np.array(centres)
    out.update(state_to_np(enc.state_dict(), "after2/"))
    out["after2/temp"] = lossf.temp.detach().numpy().copy()

    enc.eval()
    with torch.no_grad():
        Ze = enc(X, subj)
        le, losse = lossf(Y, Ze, return_logits=True)
    out["eval/Z"] = Ze.numpy().copy()
    out["eval/logits"] = le.numpy().copy()
    out["eval/loss"] = losse.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "e2e_small.npz"), **out)
    print("e2e_small: loss0 %.6f loss1 %.6f eval %.6f centres %s" %
          (out["step0/loss"], out["step1/loss"], out["eval/loss"], centres))


def gen_classifier():
    B, F, T = 24, 5, 7
    g = torch.Generator().manual_seed(5)
    Y = torch.randn(B, F, T, generator=g)
    Z = 0.12 * Y + torch.randn(B, F, T, generator=g)      # partially aligned ⇒ non-trivial ranks
    clf = ref_models.Classifier(Args())
    top1, top10 = clf(Z, Y)
    np.savez_compressed(os.path.join(HERE, "classifier.npz"), Z=Z.numpy(), Y=Y.numpy(),
                        top1=np.float64(top1), top10=np.float64(top10))
    print("classifier: top1 %.4f top10 %.4f" % (top1, top10))


def gen_spot(C, S, tag):
    D1, D2, F, K, T, B = 270, 320, 1024, 32, 360, 8
    loc = O.synthetic_positions(C, seed=0)
    P = O.seeded_params(C, S, D1, D2, F, K, seed=0, loc=loc)
    args = make_args(C, S, D1, D2, 512, K, last4layers=True)     # F forced to 1024, models.py:176
    torch.manual_seed(0)
    enc, lossf = build_reference(args, loc)
    enc.load_state_dict(P)
    X, Y, subj = O.synthetic_batch(B, C, T, F, S, seed=1234)
    enc.train()
    centre = peek_drop_centre(3, C)
    Z = enc(X, subj)
    logits, loss = lossf(Y, Z, return_logits=True)
    loss.backward()
    rng = np.random.RandomState(99)
    out = {"dims": np.array([C, S, D1, D2, F, K, T, B]), "drop_centre": np.array(centre),
           "loss": loss.detach().numpy().copy(), "logits": logits.detach().numpy().copy(),
           "grad/temp": lossf.temp.grad.numpy().copy()}
    zi = rng.randint(0, Z.numel(), size=2048)
    out["Z@idx"] = zi
    out["Z@val"] = Z.detach().reshape(-1).numpy()[zi].copy()
    out["Z@sumsq"] = np.float64((Z.detach().double() ** 2).sum().item())
    for n, p in enc.named_parameters():
        if p.grad is None:             # subject layers of subjects absent from the batch
            out[f"grad/{n}@none"] = np.array(1)
            continue
        gr = torch.view_as_real(p.grad).reshape(-1) if p.grad.is_complex() else p.grad.reshape(-1)
        idx = rng.randint(0, gr.numel(), size=min(256, gr.numel()))
        out[f"grad/{n}@idx"] = idx
        out[f"grad/{n}@val"] = gr.numpy()[idx].copy()
        out[f"grad/{n}@sumsq"] = np.float64((gr.double() ** 2).sum().item())
    sd = enc.state_dict()
    for k in range(5):
        for j in (0, 1):
            for nm in ("running_mean", "running_var"):
                key = f"conv_blocks.conv{k}.batchnorm{j}.{nm}"
                out["after1fwd/" + key] = sd[key].numpy().copy()
    # eval-mode check on the same state (running stats after one update)
    enc.eval()
    with torch.no_grad():
        Ze = enc(X, subj)
        losse = lossf(Y, Ze)
    out["eval/Z@val"] = Ze.reshape(-1).numpy()[zi].copy()
    out["eval/loss"] = losse.numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"spot_{tag}.npz"), **out)
    print(f"spot_{tag}: loss {float(loss):.6f} eval {float(losse):.6f} centre {centre}")


def gen_collate():
    """Batch collate of gwilliams2022.py:651-661, computed by the reference's own preproc_utils functions
    (baseline_correction_single + scaleAndClamp, which call sklearn's RobustScaler)."""
    from speech_decoding.utils import preproc_utils as PU
    g = torch.Generator().manual_seed(21)
    X = torch.randn(5, 7, 360, generator=g) * torch.rand(5, 7, 1, generator=g) * 4 + torch.randn(5, 7, 1, generator=g)
    X[1, 2] = 0.25                                   # constant row: zero inter-quartile range
    X[0, 0, 100], X[3, 4, 7] = 500.0, -300.0         # artefact spikes: exercise the clamp
    out = PU.scaleAndClamp(PU.baseline_correction_single(X.clone(), 60), 20, True)
    out_nc = PU.scaleAndClamp(PU.baseline_correction_single(X.clone(), 60), 20, False)
    np.savez_compressed(os.path.join(HERE, "collate.npz"), X=X.numpy(), baseline_len=np.array(60), clamp_lim=np.array(20.0),
                        out=out.numpy(), out_noclamp=out_nc.numpy())
    print("collate: max |out| %.3f (unclamped %.3f)" % (out.abs().max(), out_nc.abs().max()))


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "collate":
        gen_collate()
        sys.exit(0)
    gen_e2e_small()
    gen_classifier()
    gen_spot(208, 27, "208")
    gen_spot(60, 1, "60")
    gen_collate()
