"""Every shipped instantiation / launch mapping of the rollout kernels under an oracle test ("which test
launches which kernel").  The batch size selects the mapping (csrc/od_model_tu.inc::coop_rpw, rollout_state64):
   B <= 1024: one row (problem) per wavefront        B <= 2048: two        B <= 4096: four
   4096 < B <= 8192: four rows, `k_rollout_state_coop<., 2>` -- the 256-register build, two wavefronts per SIMD.
CPU tier: the same checks at small sizes on the host build (the mapping arithmetic, not the register allocation)."""
import pytest

import parity_checks as P


def test_rollout_mappings_emulated(oracle, emu_lib):
    P.check_rollout_instantiation(oracle, emu_lib, "cpu", 37, 6, 16, n_oracle=37, t_chain=(0, 5))


def test_rollout_eight_lane_form_emulated(oracle, emu_lib):
    """the hopper through od_coop3.h (what batches of 4097..8192 run): forced on a small batch, against the 16-lane form"""
    import torch
    import workloads as W
    x1, U = W.hopper_rollout_inputs(24, 8, seed=23, u_sigma=0.7)
    im = P.make_im("hopper", emu_lib, "cpu")
    outs = []
    for mode in (2, 3):
        im.set_cooperative(mode)
        X, G, st, it, _ = im.rollout_compact(torch.tensor(x1), torch.tensor(U))
        outs.append((X.clone(), st.clone(), it.clone()))
        D, Gs, s1, i1 = im.step_grad_compact(X[:, 3].contiguous(), torch.tensor(U[:, 3]))
        assert torch.equal(D, X[4:, 4]) and torch.equal(s1, st[3]) and torch.equal(i1, it[:, 3])       # rollout == chained steps
    assert torch.equal(outs[0][1], outs[1][1]) and (outs[0][2] == outs[1][2]).double().mean().item() > 0.99
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-8


def test_plumbing_config_callbacks_emulated(oracle, emu_lib):
    P.check_plumbing_config_callbacks(oracle, emu_lib, "cpu")


@pytest.mark.gpu
def test_rollout_8192_eight_lanes_per_problem(oracle, gpu_lib):
    """B = 8192 (BASELINE config 4 on one GPU): k_rollout_state_coop3<Coop3_hopper>, 1024 wavefronts of 8 problems; reference
    mapping: the same trajectories in two batches of 4096 through the 16-lane kernel k_rollout_state_coop<Coop_hopper>"""
    P.check_rollout_instantiation(oracle, gpu_lib, "cuda:0", 8192, 20, 4096, t_chain=(0, 7, 19), same_form=False)


@pytest.mark.gpu
def test_config4_8192_rollouts_full_horizon(oracle, gpu_lib):
    """BASELINE config 4 as stated, on one GPU: hopper, 8192 rollouts x T = 100 (k_rollout_state_coop3<Coop3_hopper>), against the
    oracle's rollout on 256 trajectories over the full horizon, chained steps at the start, middle and end, and the same
    trajectories in two batches of 4096 through the 16-lane kernel"""
    P.check_rollout_instantiation(oracle, gpu_lib, "cuda:0", 8192, 100, 4096, n_oracle=256, t_chain=(0, 49, 99), same_form=False)


@pytest.mark.gpu
def test_rollout_12000_eight_lanes_two_wavefronts_per_simd(oracle, gpu_lib):
    """B = 12 000: k_rollout_state_coop3<Coop3_hopper, 2> (the 256-register build, 8193..16 384 rollouts); reference mapping: batches
    of 6000 through k_rollout_state_coop3<Coop3_hopper, 1> -- one kernel form, two builds: identical results"""
    P.check_rollout_instantiation(oracle, gpu_lib, "cuda:0", 12000, 10, 6000, t_chain=(0, 9))


@pytest.mark.gpu
def test_rollout_4100_eight_lanes_ragged(oracle, gpu_lib):
    """B = 4100: the 8-lane form with a ragged last wavefront; reference: batches of 2050 through the 16-lane form (rpw = 4)"""
    P.check_rollout_instantiation(oracle, gpu_lib, "cuda:0", 4100, 12, 2050, t_chain=(0, 11), same_form=False)


@pytest.mark.gpu
def test_rollout_two_rows_per_wavefront(oracle, gpu_lib):
    """B = 1536: rpw = 2; reference mapping: batches of 512 (rpw = 1)"""
    P.check_rollout_instantiation(oracle, gpu_lib, "cuda:0", 1536, 20, 512, t_chain=(0, 7, 19))


@pytest.mark.gpu
def test_rollout_one_row_per_wavefront_full_horizon(oracle, gpu_lib):
    """B = 1024, T = 100: BASELINE config 4's share of one GPU (8192 rollouts over 8), rpw = 1; the same trajectories are
    then rolled out as part of a 2048 batch (rpw = 2) and of a 4096 batch (rpw = 4): identical results"""
    import numpy as np
    import torch
    import workloads as W
    B, T = 1024, 100
    im = P.check_rollout_instantiation(oracle, gpu_lib, "cuda:0", B, T, B, t_chain=(0, 37, 99))
    # the same 1024 trajectories as the first half of a 2048 batch (rpw = 2) and of a 4096 batch (rpw = 4)
    x1, U = W.hopper_rollout_inputs(B, T, seed=23, u_sigma=0.7)
    X, G, st, it, _ = im.rollout_compact(torch.tensor(x1, device="cuda:0"), torch.tensor(U, device="cuda:0"))
    X, it = X.clone(), it.clone()
    for rep in (2, 4):
        Xr, Gr, sr, ir, _ = im.rollout_compact(torch.tensor(np.tile(x1, (1, rep)), device="cuda:0"), torch.tensor(np.tile(U, (1, 1, rep)), device="cuda:0"))
        assert torch.equal(ir[:, :, :B], it) and torch.equal(Xr[:, :, :B], X), rep
        assert torch.equal(Xr[:, :, -B:], X), rep


@pytest.mark.gpu
def test_plumbing_config_callbacks(oracle, gpu_lib):
    """BASELINE config 1 on the device through od_f_host / od_fx_host / od_fu_host"""
    P.check_plumbing_config_callbacks(oracle, gpu_lib, "cuda:0")
