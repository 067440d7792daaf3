"""Measurement and profiling tools (not part of the product package)."""
