"""Scalar restatement of the reference's slice sampler, one chain at a time.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): imported by tests/test_slicesample.py as the
checker of the product's lock-step engine (desi-mcmc_amd/util/infer/slicesample.py).  The product
never imports this module.

Follows CelestePy/util/infer/slicesample.py:89-227 (`slicesample`: direction_slice :114-203 with
`acceptable` :119-131, the component-wise / random-direction drivers :213-228) statement by statement;
the only change is where the random numbers come from: every `npr.rand()` / `npr.randn()` of the
reference is a draw from ONE stream handed in by the caller (the product's per-chain stream), in the
reference's order, so that a chain of the lock-step engine can be compared with it draw for draw.
`npr.shuffle(ordering)` (:216) becomes a stable argsort of one uniform per axis, as the engine does.
Parity status: pinned by construction to the reference's control flow; the reference's own
`__main__` demo (:230-283) is not a test and pins nothing.
"""
import numpy as np


def scalar_slicesample(init_x, logprob, stream, chain, sigma=1.0, step_out=True, max_steps_out=1000, compwise=True,
                       numdir=2, doubling_step=True):
    """slicesample.py:114-228, one chain, uniforms taken from stream `chain` in the reference's order"""
    one = np.array([chain])

    def rand():
        return stream.uniform(one)[0]

    def randn():
        return stream.normal(one)[0]

    def direction_slice(direction, init_x):
        def dir_logprob(z):
            return logprob(direction * z + init_x)

        def acceptable(z, llh_s, L, U):
            while (U - L) > 1.1 * sigma:
                middle = 0.5 * (L + U)
                splits = (middle > 0 and z >= middle) or (middle <= 0 and z < middle)
                if z < middle:
                    U = middle
                else:
                    L = middle
                if splits and llh_s >= dir_logprob(U) and llh_s >= dir_logprob(L):
                    return False
            return True
        upper = sigma * rand()
        lower = upper - sigma
        llh_s = np.log(rand()) + dir_logprob(0.0)
        l_steps_out = u_steps_out = 0
        if step_out:
            if doubling_step:
                while (dir_logprob(lower) > llh_s or dir_logprob(upper) > llh_s) and (l_steps_out + u_steps_out) < max_steps_out:
                    if rand() < 0.5:
                        l_steps_out += 1
                        lower -= (upper - lower)
                    else:
                        u_steps_out += 1
                        upper += (upper - lower)
            else:
                while dir_logprob(lower) > llh_s and l_steps_out < max_steps_out:
                    l_steps_out += 1
                    lower -= sigma
                while dir_logprob(upper) > llh_s and u_steps_out < max_steps_out:
                    u_steps_out += 1
                    upper += sigma
        start_upper, start_lower = upper, lower
        while True:
            new_z = (upper - lower) * rand() + lower
            new_llh = dir_logprob(new_z)
            if new_llh > llh_s and acceptable(new_z, llh_s, start_lower, start_upper):
                break
            elif new_z < 0:
                lower = new_z
            elif new_z > 0:
                upper = new_z
            else:
                raise Exception("Slice sampler shrank to zero!")
        return new_z * direction + init_x, new_llh
    dims = init_x.shape[0]
    if compwise:
        ordering = np.argsort([rand() for _ in range(dims)], kind="stable")
        new_x = init_x.copy()
        for d in ordering:
            direction = np.zeros(dims)
            direction[d] = 1.0
            new_x, new_llh = direction_slice(direction, new_x)
    else:
        new_x = init_x
        for d in range(numdir):
            direction = np.array([randn() for _ in range(dims)])
            direction = direction / np.sqrt(np.sum(direction ** 2))
            new_x, new_llh = direction_slice(direction, new_x)
    return new_x, new_llh
