"""Re-export of synthdata/fixtures.py (the seeded input generators moved out of oracle/ so that bench.py's timed path
imports nothing from the oracle package)."""
from synthdata.fixtures import *          # noqa: F401,F403
from synthdata.fixtures import CONFIGS, _rs, load_seeded_state, renderer_inputs, seeded_param, to_torch   # noqa: F401
