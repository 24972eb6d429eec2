NESTING_LVL = 5      # reference from_msa/__init__.py:2-3
MIN_MATCH_LEN = 7
