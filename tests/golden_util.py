"""Helpers to load the committed reference vectors (tests/golden, made by oracle/gen_golden.py)."""
import json
import os

import numpy as np
import torch

import pairs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


def e2e_cases():
    return [c for c in load_manifest()["cases"] if c["kind"] in pairs.PAIR_KINDS]


def demo_cases():
    return [c for c in load_manifest()["cases"] if c["kind"] == "demo_uint8"]


def load_npz(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def case_inputs(c):
    fn = pairs.PAIR_KINDS[c["kind"]]
    im0, im1 = fn(c["B"], c["H"], c["W"], c["seed"])
    return im0, im1


def check_inputs_match(c, gold, im0, im1):
    """Inputs are regenerated from the seed; make sure they are the ones the reference saw."""
    s = gold["in_sums"]
    assert abs(im0.double().sum().item() - s[0]) < 1e-6 * max(1.0, abs(s[0]))
    assert abs(im1.double().sum().item() - s[1]) < 1e-6 * max(1.0, abs(s[1]))


def compare_e2e(out, gold, step, tol, tol_flow=None):
    """max|d| of every stored tensor; returns dict of errors (asserts on tol)."""
    tol_flow = tol if tol_flow is None else tol_flow
    errs = {}
    def sub(t):
        return t[..., ::step, ::step].detach().float().cpu().numpy()
    for key, val, tl in (("I_t", out["I_t"], tol), ("im_t0", out["im_t_list"][0], tol),
                         ("opt_flow_0", out["opt_flow_0"], tol_flow), ("opt_flow_1", out["opt_flow_1"], tol_flow),
                         ("occ_mask1", out["occ_mask1"], tol), ("I_t_0", out["I_t_0"], tol), ("I_t_1", out["I_t_1"], tol)):
        errs[key] = float(np.abs(sub(val) - gold[key]).max())
        assert errs[key] <= tl, f"{key}: max|d| {errs[key]:.3e} > {tl}"
    for key, val in (("im_t_coarse", out["im_t_list"][-1]), ("im0_warped_coarse", out["im0_warped_list"][-1])):
        errs[key] = float(np.abs(val.detach().float().cpu().numpy() - gold[key]).max())
        assert errs[key] <= tol, f"{key}: max|d| {errs[key]:.3e} > {tol}"
    return errs
