"""The arithmetic behind the filter kernel's rotated tiles and check points
(cuking_amd/csrc/king_common.h `phase_step`, king_filter.hip `sample_stats_kernel`,
`steps_of`, `prefix_u_of`), restated in Python and checked exhaustively for the bitset
lengths the library accepts: the k-steps of a bitset fall into 128 phases, the statistics
kernel finds a k-step's phase by one division, and the k-steps / per-sample counts of a run of
phases that starts anywhere and goes around the end of the sites add up."""
import numpy as np
import pytest

PHASES = 128


def phase_step(all_steps, x):
    return all_steps * x // PHASES


def phase_of(all_steps, step):
    """What sample_stats_kernel stores in its LDS table."""
    return min((PHASES * (step + 1) - 1) // all_steps, PHASES - 1)


@pytest.mark.parametrize("all_steps", list(range(1, 300)) + [391, 586, 782, 1000, 4095, 4096, 16384])
def test_every_k_step_lies_in_the_phase_the_statistics_kernel_names(all_steps):
    steps = np.arange(all_steps)
    x = np.array([phase_of(all_steps, int(s)) for s in steps])
    lo = np.array([phase_step(all_steps, int(v)) for v in x])
    hi = np.array([phase_step(all_steps, int(v) + 1) for v in x])
    assert np.all(lo <= steps) and np.all(steps < hi)
    assert phase_step(all_steps, 0) == 0 and phase_step(all_steps, PHASES) == all_steps


@pytest.mark.parametrize("all_steps", [25, 128, 391, 782])
def test_a_run_of_phases_around_the_end_covers_its_k_steps_once(all_steps):
    rng = np.random.default_rng(all_steps)
    u = rng.integers(-3, 200, size=all_steps)          # |Y| - |M| of one sample per k-step
    cum = np.concatenate([[0], np.cumsum(u)])           # ... cumulative, at every k-step
    total = int(cum[-1])

    def cum_at(x):                                       # what the workspace holds: boundaries 1 .. 127
        return int(cum[phase_step(all_steps, x)])

    for phase in range(PHASES):
        k0 = phase_step(all_steps, phase)
        for share in (8, 32, 50, 56, 61, 62, 64):       # in 64ths: two phases each
            hi = phase + 2 * share
            # king_filter.hip steps_of
            steps = (phase_step(all_steps, hi) - k0 if hi <= PHASES
                     else (all_steps - k0) + phase_step(all_steps, hi - PHASES))
            # king_filter.hip prefix_u_of
            xb = hi - PHASES if hi > PHASES else hi
            got = (total if hi > PHASES else 0) + (total if xb == PHASES else cum_at(xb)) - cum_at(phase)
            # the k-steps the tile has made by then: k0, k0 + 1, ... around the end
            walked = [(k0 + s) % all_steps for s in range(steps)]
            assert len(set(walked)) == steps <= all_steps
            assert got == int(u[walked].sum()), (phase, share)


@pytest.mark.parametrize("all_steps", [25, 130, 391])
def test_request_addresses_go_around_the_end_of_the_sites(all_steps):
    """king_filter.hip addr_of / addr_next: inside a segment [seg_first, seg_end) of a rotated
    tile the request for k-step `step` of the segment names k-step seg_k + step of the
    bitset, all_steps back from `seg_wrap` on -- i.e. (k0 + seg_first + step) mod all_steps --,
    and steps beyond the segment repeat its last one."""
    for phase in range(0, PHASES, 7):
        k0 = phase_step(all_steps, phase)
        wrap = all_steps - k0 if k0 != 0 else 0
        for cuts in ([all_steps], [all_steps // 8, all_steps * 56 // 64, all_steps]):
            seg_first = 0
            for seg_end in cuts:
                if seg_end <= seg_first:
                    continue
                seg_steps = seg_end - seg_first
                seg_k = seg_first - wrap if wrap != 0 and seg_first >= wrap else k0 + seg_first
                seg_wrap = wrap - seg_first if seg_first < wrap < seg_end else 1 << 32
                # addr_of for the first stages, addr_next from there on
                pos = None
                for step in range(seg_steps + 4):
                    if step < 4:
                        s = min(step, seg_steps - 1)
                        pos = seg_k + s - (all_steps if s >= seg_wrap else 0)
                    else:
                        pos = pos + (1 if step < seg_steps else 0) - (all_steps if step == seg_wrap else 0)
                    want = (k0 + seg_first + min(step, seg_steps - 1)) % all_steps
                    assert pos == want, (phase, seg_first, seg_end, step)
                seg_first = seg_end
