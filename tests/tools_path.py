"""Puts tools/ on sys.path so tests can import the seeded model writers as modules."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
