"""The parts of bench.py (repo root): constants and small helpers (common), the rank control plane (control), the timed region and the
JSON line (headline), the other BASELINE readings (paths), the live PMC passes (traffic), the sharded-SRS block (sharded), and the two
modules that may touch oracle/: the after-the-timer checkers (checks) and the CPU baseline's worker pool (cpu_pool)."""
