"""Stand-in for IPython (absent): the reference only imports display() for pretty printing."""
