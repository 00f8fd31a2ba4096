"""Re-export of synthdata/synth.py (see oracle/fixtures.py)."""
from synthdata.synth import *             # noqa: F401,F403
