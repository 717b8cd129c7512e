"""CPU restatement of the learner's parameter update — TEST INFRASTRUCTURE ONLY (tests/ and nothing else import it;
parl_amd never does).

What the reference does per update (parl/algorithms/torch/a2c.py:76-78; the Paddle IMPALA configures the same pair as
`Adam(learning_rate, grad_clip=ClipGradByGlobalNorm(40))`, parl/algorithms/paddle/impala.py:113-117):

    torch.nn.utils.clip_grad_norm_(parameters, max_norm)      # g *= min(1, max_norm / (||g||_2 over ALL tensors + 1e-6))
    optimizer.step()                                          # torch.optim.Adam, no weight decay / amsgrad

restated in numpy float32 with the operation order of `parlhip_clip_adam_f32` (include/parl_hip.h): step += 1; the
scalars of the step formed in float64 from the Python-float hyper-parameters (1 - beta, the bias corrections,
lr / (1 - beta1^step), sqrt(1 - beta2^step)) and rounded to float32 once; the element arithmetic in float32:
    m = m + (1 - beta1) (g - m);  v = beta2 v + (1 - beta2) g g;  p = p - step_size * (m / (sqrt(v) / bc2_sqrt + eps)).
The global norm is accumulated in float64 here (the kernel: float32 partial sums of 2,048-element chunks in a fixed
order) — the one place where the two differ by more than an ulp of the inputs, and by far less than the tolerance
the clip factor is compared with.

PINNED (tests/test_optim_oracle.py, CPU): against torch's own clip_grad_norm_ + torch.optim.Adam on the host — the
functions the reference calls — over several steps with a changing learning rate, clip active and inactive."""
import numpy as np


def clip_adam_step(params, grads, exp_avg, exp_avg_sq, steps, lr, beta1=0.9, beta2=0.999, eps=1e-8, max_norm=40.0):
    """In place on lists of float32 numpy arrays (`steps`: list of python / numpy scalars, replaced in the list).
    Returns the global gradient norm before clipping (float)."""
    total = 0.0
    for g in grads:
        total += float(np.sum(g.astype(np.float64) ** 2))
    norm = np.float32(np.sqrt(total))
    clip = np.float32(max_norm) / (norm + np.float32(1e-6))
    clip = np.float32(min(clip, np.float32(1.0)))
    w1, w2, b2f = np.float32(1.0 - beta1), np.float32(1.0 - beta2), np.float32(beta2)
    for i in range(len(params)):
        steps[i] = float(steps[i]) + 1.0
        t = steps[i]
        bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
        step_size = np.float32(float(lr) / bc1)
        bc2_sqrt = np.float32(np.sqrt(bc2))
        g = grads[i]
        g *= clip                                              # clipped in place, as clip_grad_norm_ does
        m, v, p = exp_avg[i], exp_avg_sq[i], params[i]
        m += w1 * (g - m)
        v *= b2f
        v += (w2 * g) * g
        denom = np.sqrt(v) / bc2_sqrt + np.float32(eps)
        p -= step_size * (m / denom)
    return float(norm)
