python -m pytest tests/test_analysis_gpu.py -x -q -k "flat_cells or last_resort or default_options or quality_arm" 2>&1 | tail -8
python -m pytest tests/test_oracle_fixtures_gpu.py -x -q -k "uvsphere" 2>&1 | tail -4
