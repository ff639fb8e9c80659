"""The pieces of bench.py (repo root): one module per benchmark family.  Nothing here is imported by the product."""
