"""CPU oracle for the ray-march path — TEST INFRASTRUCTURE ONLY (see kpnerf_oracle.c)."""
