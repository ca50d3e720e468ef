"""Setting `deepmimic_dppo_base` (values and reference citations: _table.py)."""
from ._table import publish

publish(globals(), 'dppo', pfpn=False)
