"""The parity CHECKER is frozen ("What's weak" 4 / ADVICE): oracle/parity.py's acceptance constants and the source of its
criterion functions are pinned by tests/golden/parity_protocol.json.  A change to either fails here until the golden file is updated IN
THE SAME COMMIT with a written rationale (its `history` list) -- the constants may not move because a hardware run failed."""
import hashlib
import inspect
import json
import os

from oracle import parity

HERE = os.path.dirname(os.path.abspath(__file__))
FROZEN = os.path.join(HERE, 'golden', 'parity_protocol.json')
FUNCS = ('_align', '_rel_errors', 'sample_protocol', 'truth_protocol', 'image_protocol')


def current():
    consts = dict(THRESH2=parity.THRESH2, EPS=parity.EPS, FLOOR_SIGMA=parity.FLOOR_SIGMA, FLOOR_RGB=parity.FLOOR_RGB,
                  QUANTILES=list(parity.QUANTILES), MAX_FACTOR=parity.MAX_FACTOR, MIN_TAIL=parity.MIN_TAIL)
    code = {f: hashlib.sha256(inspect.getsource(getattr(parity, f)).encode()).hexdigest()[:16] for f in FUNCS}
    return dict(constants=consts, code_sha256_16=code)


def test_parity_protocol_is_frozen():
    frozen = json.load(open(FROZEN))
    cur = current()
    assert cur['constants'] == frozen['constants'], 'oracle/parity.py constants changed: update tests/golden/parity_protocol.json with a rationale'
    assert cur['code_sha256_16'] == frozen['code_sha256_16'], 'oracle/parity.py criterion code changed: update tests/golden/parity_protocol.json with a rationale'
    assert frozen['history'] and all('why' in h for h in frozen['history'])


if __name__ == '__main__':          # python tests/test_parity_frozen.py  -> prints the current fingerprint (to paste into the golden file)
    print(json.dumps(current(), indent=1))
