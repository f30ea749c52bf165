"""CPU oracle (TEST INFRASTRUCTURE ONLY -- see whisper_oracle.cpp header).
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
