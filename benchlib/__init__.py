"""Parts of bench.py (the measurement contract): headline.py (config 2 / 4 line), c3.py, c5.py, objects.py (bounded objects of the
other configs in the default line), lanes.py (what is timed), roofline.py (HIP events + live counter passes), baselines.py (CPU oracle /
torch eager legs), launch.py (N ranks).  Only bench.py is an entry point."""
