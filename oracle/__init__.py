"""TEST INFRASTRUCTURE.  CPU restatement (oracle) of the reference's Hamilton-product path.

Nothing under the product package may import this package; see oracle/qk_oracle.c.
"""
