"""Model builders for bench.py / smoke / parity runs on machines where the
reference checkout (and therefore its model/*/network.py) is not present.
Same architecture, attribute names (= state-dict keys) and construction order
as the reference file each cites; written against the furnace surface only."""
import os
import sys

_FURNACE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "furnace")


def ensure_furnace_on_path():
    """What the reference's config.py:49-54 does with <TorchSeg>/furnace."""
    if _FURNACE not in sys.path:
        sys.path.insert(0, _FURNACE)
