"""Parity, plumbing and property tests of the MI355X super-resolution hot path (CPU: oracle and host logic; `-m gpu`: the HIP path through the C ABI)."""
