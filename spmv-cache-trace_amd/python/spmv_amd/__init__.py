"""Host-side Python plumbing for the MI355X SpMV path (ctypes over the C ABI)."""
