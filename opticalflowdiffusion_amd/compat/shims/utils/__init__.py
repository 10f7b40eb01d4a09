"""The `utils` package main.py:9 and flow_diffuser.py:12-13 import; the reference repository does not contain it."""
