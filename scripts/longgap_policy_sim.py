"""What decoupling the four trajectories of a wavefront could buy on the long-gap grid (1 .. 10 Runge-Kutta steps per
interval, independent draws).  Cost model of the sixteen-lane grid kernels: one loop iteration costs 1 if any
row takes a Runge-Kutta step, + u if the measurement update is executed (u = update / step in issued instructions: about 0.5 for the
in-grid update of filter_lpe_kernel); rows are lanes of one wavefront, so a phase is paid for by all rows whenever any row runs it.
    lockstep: today's kernel -- every interval costs the longest of its four rows, one update per interval
    eager:    every row keeps its own observation index; the update runs whenever any row has reached an observation
    thr(n,w): ... only when n rows wait at an observation, or one has waited w iterations, or nothing else is left to do
python3 scripts/longgap_policy_sim.py"""
import numpy as np

T = 1000


def sim(rng, policy, u, thr=2, maxwait=3, reps=100):
    tot = 0.0
    for _ in range(reps):
        steps = rng.integers(1, 11, size=(4, T))
        k = np.zeros(4, int)
        rem = np.zeros(4, int)
        wait = np.zeros(4, int)
        done = np.zeros(4, bool)
        cost = 0.0
        while not done.all():
            at_obs = (rem == 0) & ~done
            stepping = (rem > 0) & ~done
            if policy == "eager":
                upd = at_obs.any()
            else:
                upd = (at_obs.sum() >= thr) or (at_obs.any() and not stepping.any()) or (wait[at_obs].max(initial=0) >= maxwait)
            if stepping.any():
                cost += 1.0
                rem[stepping] -= 1
            if upd:
                cost += u
                for r in np.where(at_obs)[0]:
                    k[r] += 1
                    wait[r] = 0
                    if k[r] >= T:
                        done[r] = True
                    else:
                        rem[r] = steps[r, k[r]]
            else:
                wait[at_obs] += 1
            tot += 0.0
        tot += cost
    return tot / reps


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    print("u      lockstep   eager   thr(2,3)  thr(2,2)  thr(3,4)   best / lockstep")
    for u in (0.3, 0.5, 0.8):
        lock = np.mean([rng.integers(1, 11, size=(4, T)).max(0).sum() + T * u for _ in range(100)])
        r = [sim(rng, "eager", u), sim(rng, "thr", u, 2, 3), sim(rng, "thr", u, 2, 2), sim(rng, "thr", u, 3, 4)]
        print(f"{u:3.1f}  {lock:9.0f}  " + "  ".join(f"{x:8.0f}" for x in r) + f"   {min(r) / lock:.3f}")
