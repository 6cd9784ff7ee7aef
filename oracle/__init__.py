"""CPU oracle for the fnft_nsev hot path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
fnft_amd/ (tests/test_no_oracle_in_product.py enforces that).
"""
from .oracle import Oracle, load_oracle, build_oracle  # noqa: F401
