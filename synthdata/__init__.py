"""Synthetic inputs: seeded data generators (synthetic SMPL body, poses, cameras, rays, feature tables, parameter values)
shared by the oracle, the tests and bench.py.  No part of the rendering algorithm lives here and nothing here is the
oracle: `oracle/` re-exports these modules under their old names for the tests."""
