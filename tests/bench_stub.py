"""A planner stand-in for dry runs of bench.py's JOB logic on CPU (TDMPC2_BENCH_STUB=1, tests/test_bench_contract.py): the
methods bench.py calls on NativePlanner, no kernels, a fixed fake step time.  TEST INFRASTRUCTURE ONLY -- never a measurement."""
import time


class StubPlanner:
    STEP_S = 0.004

    def __init__(self, cfg, iterations, device, max_envs=1, path=0, precision=0):
        self.cfg, self.iterations, self.device = cfg, iterations, device
        self.path = path or (1 if (cfg.latent_dim, cfg.mlp_dim) == (512, 512) else 2)
        self.precision = precision or 2
        self.device_bytes = 0
        self._prof, self._launches = 0, 0

    def bind_state_dict(self, sd):
        pass

    def plan(self, z0, disc_pow, prev_mean, t0, eval_mode=False, task_emb=None, act_mask=None, tape=None, seed=0, out=None,
             debug=False):
        time.sleep(self.STEP_S)
        out.fill_(0.01 * (int(seed) % 7))
        if self._prof:
            self._launches = min(self._launches + self.iterations, self._prof)
        return out

    def set_profiling(self, max_launches):
        self._prof, self._launches = int(max_launches), 0

    def profile_read(self):
        n, self._launches = self._launches, 0
        return 1e3 * self.STEP_S * n / max(self.iterations, 1), n

    def take_fault(self):
        return 0

    def set_rows_per_workgroup(self, rows):
        pass

    def set_fold_refit(self, mode):
        pass

    def close(self):
        pass
