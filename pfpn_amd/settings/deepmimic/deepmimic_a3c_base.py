"""Setting `deepmimic_a3c_base` (values and reference citations: _table.py)."""
from ._table import publish

publish(globals(), 'a3c', pfpn=False)
