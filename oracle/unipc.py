"""ORACLE (test infrastructure).  Literal 0-d-tensor restatement of diffusers==0.33.1
`UniPCMultistepScheduler` for the configuration built at /root/reference/inference_t23d.py:65-70
(flow_prediction, use_flow_sigmas, flow_shift, solver_order=2, bh2, predict_x0, lower_order_final,
final_sigmas_type="zero").  The scheduler class is in the un-vendored diffusers wheel -> PARITY UNPINNED;
guarded by known-answer tests (sigma table, first-order step = DDIM-like closed form, sigma=0 terminal step)."""
from __future__ import annotations

import numpy as np
import torch


class OracleUniPC:
    def __init__(self, num_train_timesteps=1000, flow_shift=1.0, solver_order=2, solver_type="bh2"):
        self.T, self.shift, self.order_max, self.solver_type = num_train_timesteps, flow_shift, solver_order, solver_type

    def set_timesteps(self, n):
        alphas = np.linspace(1, 1 / self.T, n + 1)
        s = 1.0 - alphas
        s = np.flip(self.shift * s / (1 + (self.shift - 1) * s))[:-1].copy()
        self.timesteps = torch.from_numpy((s * self.T).copy()).to(torch.int64)
        self.sigmas = torch.from_numpy(np.concatenate([s, [0.0]]).astype(np.float32))
        self.model_outputs = [None] * self.order_max
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = 0
        self.this_order = 1

    @staticmethod
    def _as(sigma):
        return 1 - sigma, sigma

    def _rb(self, rks, hh, order):
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.solver_type == "bh1" else torch.expm1(hh)
        R, b, f = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * f / B_h)
            f *= i + 1
            h_phi_k = h_phi_k / hh - 1 / f
        return torch.stack(R), torch.tensor(b), h_phi_1, B_h

    def _p(self, x, order):
        m0 = self.model_outputs[-1]
        sigma_t, sigma_s0 = self.sigmas[self.step_index + 1], self.sigmas[self.step_index]
        alpha_t, sigma_t = self._as(sigma_t)
        alpha_s0, sigma_s0 = self._as(sigma_s0)
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks, D1s = [], []
        for i in range(1, order):
            si = self.step_index - i
            mi = self.model_outputs[-(i + 1)]
            a, s = self._as(self.sigmas[si])
            rk = ((torch.log(a) - torch.log(s)) - lambda_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        R, b, h_phi_1, B_h = self._rb(rks, -h, order)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        if D1s:
            D1s = torch.stack(D1s, dim=1)
            rhos_p = torch.tensor([0.5], dtype=x.dtype) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1]).to(x.dtype)
            pred_res = torch.einsum("k,bkc...->bc...", rhos_p, D1s)
        else:
            pred_res = 0
        return (x_t_ - alpha_t * B_h * pred_res).to(x.dtype)

    def _c(self, model_t, x, order):
        m0 = self.model_outputs[-1]
        sigma_t, sigma_s0 = self.sigmas[self.step_index], self.sigmas[self.step_index - 1]
        alpha_t, sigma_t = self._as(sigma_t)
        alpha_s0, sigma_s0 = self._as(sigma_s0)
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks, D1s = [], []
        for i in range(1, order):
            si = self.step_index - (i + 1)
            mi = self.model_outputs[-(i + 1)]
            a, s = self._as(self.sigmas[si])
            rk = ((torch.log(a) - torch.log(s)) - lambda_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        R, b, h_phi_1, B_h = self._rb(rks, -h, order)
        D1s = torch.stack(D1s, dim=1) if D1s else None
        rhos_c = torch.tensor([0.5], dtype=x.dtype) if order == 1 else torch.linalg.solve(R, b).to(x.dtype)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        corr_res = torch.einsum("k,bkc...->bc...", rhos_c[:-1], D1s) if D1s is not None else 0
        D1_t = model_t - m0
        return (x_t_ - alpha_t * B_h * (corr_res + rhos_c[-1] * D1_t)).to(x.dtype)

    def step(self, model_output, sample):
        si = self.step_index
        use_corrector = si > 0 and self.last_sample is not None
        m = sample - self.sigmas[si] * model_output
        if use_corrector:
            sample = self._c(m, self.last_sample, self.this_order)
        for i in range(self.order_max - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = m
        this_order = min(self.order_max, len(self.timesteps) - si)
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._p(sample, self.this_order)
        if self.lower_order_nums < self.order_max:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev
