"""Per-case parity expectations.

The north-star tolerance is <= 1 LSB per 8-bit channel, and that is what every case is held
to unless a committed expectation file says otherwise for that exact case.  A filter with
discontinuities (modulo, comparisons on libm results, noise lattice look-ups) can turn one
float ulp of difference between two libms into a different branch; such cases are listed in
``tests/golden/expected_<suite>.json`` with the measured ``[max, n_gt1]`` (largest channel
difference, number of channel values that differ by more than 1), and the test asserts the
measurement does not exceed the record.  There is no blanket allowance: a new case, or a
listed case that got worse, fails.

Every check also appends the measurement to ``gpurun_out/parity_<suite>.json`` (scratch, merged
back from the GPU box) so the expectation files can be regenerated with
``MM_RECORD_EXPECTATIONS=1`` (records instead of asserting) and reviewed in the diff.
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_reports = {}


class Expectations:
    def __init__(self, suite):
        self.suite = suite
        self.path = os.path.join(ROOT, "tests", "golden", "expected_%s.json" % suite)
        self.data = json.load(open(self.path)) if os.path.exists(self.path) else {}
        self.record_only = bool(os.environ.get("MM_RECORD_EXPECTATIONS"))

    def check(self, case, mx, n_diff, n_gt1, total=None, default=(1, 0)):
        """Asserts (mx, n_gt1) <= the record for `case` (default: <= 1 LSB, nothing beyond)."""
        rep = _reports.setdefault(self.suite, {})
        rep[case] = {"max": mx, "n_diff": n_diff, "n_gt1": n_gt1, "total": total}
        out = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_%s.json" % self.suite), "w") as f:
                json.dump(rep, f, indent=1, sort_keys=True)
        except OSError:
            pass
        if self.record_only:
            return
        exp_max, exp_gt1 = self.data.get(case, default)
        assert mx <= exp_max and n_gt1 <= exp_gt1, (
            "%s / %s: max diff %d (%d values differ, %d by more than 1); allowed: max %d, %d beyond 1 LSB"
            % (self.suite, case, mx, n_diff, n_gt1, exp_max, exp_gt1))
