"""Setting `deepmimic_impala_base` (values and reference citations: _table.py)."""
from ._table import publish

publish(globals(), 'impala', pfpn=False)
