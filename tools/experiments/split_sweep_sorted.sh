for k in 3 4 5 6 7 9 12 16; do echo "== RMD_SAMPLE_SPLIT=$k"; RMD_SAMPLE_SPLIT=$k python3 tools/quick_time.py C2 500 | grep kernel | sed -E 's/.*kernel ([0-9.]+) ms.*/\1/' | sort -n | head -1; done
