"""Setting `deepmimic_a3c_pfpn` (values and reference citations: _table.py)."""
from ._table import publish

publish(globals(), 'a3c', pfpn=True)
