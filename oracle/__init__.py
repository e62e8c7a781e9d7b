"""CPU oracle for the tet-sphere geometry energy -- test infrastructure only.

Nothing under this package is imported by the product (``tssplat_amd``).
"""
