"""Test-infrastructure oracle (CPU).  Never imported by the product path."""
