#!/usr/bin/env python3
"""CLI of hlynr_intercept_amd/hotcheck.py: python tools/check_hot_words.py [libhlx.so | listing.s]"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runpy.run_module("hlynr_intercept_amd.hotcheck", run_name="__main__")
