"""TEST INFRASTRUCTURE ONLY -- see oracle/np_oracle.py.  Nothing under
feinsum_amd/ may import this package."""
