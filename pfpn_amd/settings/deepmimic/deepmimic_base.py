"""Setting `deepmimic_base` (values and reference citations: _table.py)."""
from ._table import publish

publish(globals(), None, pfpn=False)
