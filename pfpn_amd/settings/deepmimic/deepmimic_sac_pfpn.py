"""Setting `deepmimic_sac_pfpn` (values and reference citations: _table.py)."""
from ._table import publish

publish(globals(), 'sac', pfpn=True)
