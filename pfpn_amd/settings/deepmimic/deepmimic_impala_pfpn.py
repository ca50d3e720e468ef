"""Setting `deepmimic_impala_pfpn` (values and reference citations: _table.py)."""
from ._table import publish

publish(globals(), 'impala', pfpn=True)
