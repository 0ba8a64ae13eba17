"""Mitsuba 0.6 / Mitsuba 3 plugin adapters (C++) over the C ABI; see build.py and INTEGRATION.md."""
