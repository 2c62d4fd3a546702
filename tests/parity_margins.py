"""Measured parity margins of the -m gpu tests.  With DA_PARITY_MARGINS=<path> every recorded case is (re)written there as
JSON next to the tolerance it was asserted against (committed per round as profiles/rNN_parity_margins.json)."""
import json
import os

MARGINS = {}
TOLERANCES = {}


def record(case, tolerances=None, **kv):
    MARGINS.setdefault(case, {}).update(kv)
    if tolerances:
        TOLERANCES[case] = dict(tolerances)
    path = os.environ.get('DA_PARITY_MARGINS')
    if path:
        os.makedirs(os.path.dirname(os.path.abspath(path)) or '.', exist_ok=True)
        with open(path, 'w') as f:
            json.dump({'tolerances': TOLERANCES, 'measured': MARGINS}, f, indent=1, sort_keys=True)
