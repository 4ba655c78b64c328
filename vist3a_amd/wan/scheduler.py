"""UniPC multistep scheduler for flow-matching sigmas — host logic of the denoise loop.

Same constructor / `set_timesteps` / `step` surface as the object the reference builds at
/root/reference/inference_t23d.py:65-70:
    UniPCMultistepScheduler(prediction_type="flow_prediction", num_train_timesteps=1000,
                            use_flow_sigmas=True, flow_shift=args.flow_shift)
(diffusers==0.33.1 defaults: solver_order=2, solver_type="bh2", predict_x0=True, lower_order_final=True,
final_sigmas_type="zero", no thresholding).  The per-step update is a handful of axpy's on a 262 144-element
latent: the coefficients are computed here on the host in float32 (as the reference does with 0-d float32
tensors) and applied with a couple of fused device ops — SURVEY.md §8 row A1 ("negligible; keep in PyTorch").
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch


class UniPCMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 use_flow_sigmas: bool = True, flow_shift: float = 1.0, solver_type: str = "bh2",
                 predict_x0: bool = True, lower_order_final: bool = True, final_sigmas_type: str = "zero",
                 disable_corrector: Optional[List[int]] = None):
        if prediction_type != "flow_prediction" or not use_flow_sigmas:
            raise NotImplementedError("only the flow_prediction / use_flow_sigmas configuration of the VIST3A path is implemented")
        if solver_type not in ("bh1", "bh2"):
            raise NotImplementedError(solver_type)
        if not predict_x0:
            raise NotImplementedError("predict_x0=False")
        self.num_train_timesteps = num_train_timesteps
        self.solver_order = solver_order
        self.flow_shift = flow_shift
        self.solver_type = solver_type
        self.lower_order_final = lower_order_final
        self.final_sigmas_type = final_sigmas_type
        self.disable_corrector = list(disable_corrector or [])
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.sigmas = None
        self._reset()

    def _reset(self):
        self.model_outputs = [None] * self.solver_order
        self.timestep_list = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = 1
        self._step_index = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: int, device=None):
        alphas = np.linspace(1, 1 / self.num_train_timesteps, num_inference_steps + 1)
        sigmas = 1.0 - alphas
        sigmas = np.flip(self.flow_shift * sigmas / (1 + (self.flow_shift - 1) * sigmas))[:-1].copy()
        timesteps = (sigmas * self.num_train_timesteps).copy()
        if self.final_sigmas_type == "sigma_min":
            sigma_last = sigmas[-1]
        elif self.final_sigmas_type == "zero":
            sigma_last = 0
        else:
            raise ValueError(self.final_sigmas_type)
        sigmas = np.concatenate([sigmas, [sigma_last]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)  # float32, host
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)  # truncation, as upstream
        self.num_inference_steps = len(timesteps)
        self._reset()

    def scale_model_input(self, sample, *a, **k):
        return sample

    # all coefficient math on 0-d float32 host tensors
    @staticmethod
    def _lam(sigma: torch.Tensor) -> torch.Tensor:
        return torch.log(1 - sigma) - torch.log(sigma)

    def _coeffs(self, order: int, s_t: torch.Tensor, s_0: torch.Tensor, prev_sigmas: List[torch.Tensor]):
        lam_t, lam_0 = self._lam(s_t), self._lam(s_0)
        h = lam_t - lam_0
        rks = [(self._lam(s) - lam_0) / h for s in prev_sigmas]
        rks.append(torch.tensor(1.0))
        rks = torch.stack([torch.as_tensor(r, dtype=torch.float32) for r in rks])
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.solver_type == "bh1" else torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return rks, torch.stack(R), torch.stack(b), h_phi_1, B_h

    def _predict(self, sample, order):
        si = self._step_index
        s_t, s_0 = self.sigmas[si + 1], self.sigmas[si]
        prev = [self.sigmas[si - i] for i in range(1, order)]
        rks, R, b, h_phi_1, B_h = self._coeffs(order, s_t, s_0, prev)
        m0 = self.model_outputs[-1]
        alpha_t = 1 - s_t
        x_t = (s_t / s_0).item() * sample - (alpha_t * h_phi_1).item() * m0
        if order > 1:
            if order == 2:
                rhos = torch.tensor([0.5])
            else:
                rhos = torch.linalg.solve(R[:-1, :-1], b[:-1])
            pred = 0
            for i in range(1, order):
                D1 = (self.model_outputs[-(i + 1)] - m0) / rks[i - 1].item()
                pred = pred + rhos[i - 1].item() * D1
            x_t = x_t - (alpha_t * B_h).item() * pred
        return x_t.to(sample.dtype)

    def _correct(self, this_m, last_sample, order):
        si = self._step_index
        s_t, s_0 = self.sigmas[si], self.sigmas[si - 1]
        prev = [self.sigmas[si - (i + 1)] for i in range(1, order)]
        rks, R, b, h_phi_1, B_h = self._coeffs(order, s_t, s_0, prev)
        m0 = self.model_outputs[-1]
        alpha_t = 1 - s_t
        x_t = (s_t / s_0).item() * last_sample - (alpha_t * h_phi_1).item() * m0
        if order == 1:
            rhos = torch.tensor([0.5])
        else:
            rhos = torch.linalg.solve(R, b)
        corr = 0
        for i in range(1, order):
            D1 = (self.model_outputs[-(i + 1)] - m0) / rks[i - 1].item()
            corr = corr + rhos[i - 1].item() * D1
        x_t = x_t - (alpha_t * B_h).item() * (corr + rhos[-1].item() * (this_m - m0))
        return x_t.to(last_sample.dtype)

    def plan_step(self) -> dict:
        """Host half of `step` for the fused device kernel (ops.unipc_cfg_step / csrc/denoise_step.hip): the same control flow and the
        same fp32 coefficient arithmetic as `step` / `_correct` / `_predict`, returned as plain floats; advances the step counters but
        touches no tensor (the kernel keeps sample, last_sample and the two x0 predictions on the device).  Do not mix with `step`
        inside one schedule."""
        if self.sigmas is None:
            raise ValueError("call set_timesteps first")
        if self._step_index is None:
            self._step_index = 0
            self._planned_last = False
        si = self._step_index
        c = dict(sigma=float(self.sigmas[si]), corr_order=0, cc1=0.0, cc2=0.0, cc3=0.0, c_rho_last=0.0, c_rho0=0.0, c_inv_rk=0.0,
                 pred_order=1, pc1=0.0, pc2=0.0, pc3=0.0, p_rho0=0.5, p_inv_rk=0.0)
        inv32 = lambda r: float(np.float32(1.0) / np.float32(r))   # ATen's CUDA `tensor / scalar` = tensor * (1 / scalar) in fp32
        if si > 0 and (si - 1) not in self.disable_corrector and self._planned_last:
            order = self.this_order
            s_t, s_0 = self.sigmas[si], self.sigmas[si - 1]
            rks, R, b, h_phi_1, B_h = self._coeffs(order, s_t, s_0, [self.sigmas[si - (i + 1)] for i in range(1, order)])
            alpha_t = 1 - s_t
            rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
            c.update(corr_order=order, cc1=(s_t / s_0).item(), cc2=(alpha_t * h_phi_1).item(), cc3=(alpha_t * B_h).item(),
                     c_rho_last=rhos[-1].item())
            if order == 2:
                c.update(c_rho0=rhos[0].item(), c_inv_rk=inv32(rks[0].item()))
            elif order > 2:
                raise NotImplementedError("fused step: solver_order <= 2")
        this_order = min(self.solver_order, len(self.timesteps) - si) if self.lower_order_final else self.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        order = self.this_order
        s_t, s_0 = self.sigmas[si + 1], self.sigmas[si]
        rks, R, b, h_phi_1, B_h = self._coeffs(order, s_t, s_0, [self.sigmas[si - i] for i in range(1, order)])
        alpha_t = 1 - s_t
        c.update(pred_order=order, pc1=(s_t / s_0).item(), pc2=(alpha_t * h_phi_1).item())
        if order == 2:
            c.update(pc3=(alpha_t * B_h).item(), p_rho0=0.5, p_inv_rk=inv32(rks[0].item()))
        elif order > 2:
            raise NotImplementedError("fused step: solver_order <= 2")
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        self._planned_last = True
        return c

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = False):
        if self.sigmas is None:
            raise ValueError("call set_timesteps first")
        if self._step_index is None:
            self._step_index = 0
        si = self._step_index
        use_corrector = si > 0 and (si - 1) not in self.disable_corrector and self.last_sample is not None
        # flow prediction -> x0.  sigma is a 0-d float32 host tensor: ATen multiplies the bf16 model output by the fp32 scalar in fp32 and
        # rounds the product to bf16 (measured on MI355X, round 3: `sig * v == bf16(float(v) * sig)` bit for bit; the scalar is NOT
        # rounded to bf16 first) - the same as the reference's scheduler, and what csrc/denoise_step.hip reproduces.
        m = sample - self.sigmas[si] * model_output
        if use_corrector:
            sample = self._correct(m, self.last_sample, self.this_order)
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
            self.timestep_list[i] = self.timestep_list[i + 1]
        self.model_outputs[-1] = m
        self.timestep_list[-1] = timestep
        if self.lower_order_final:
            this_order = min(self.solver_order, len(self.timesteps) - si)
        else:
            this_order = self.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._predict(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return (prev,)
