"""Setting `deepmimic_dppo_pfpn` (values and reference citations: _table.py)."""
from ._table import publish

publish(globals(), 'dppo', pfpn=True)
