"""Decode (paged + rotary, fp16 KV) across batch sizes / GQA ratios: is split-KV filling the chip?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_decode
for B in (1, 4, 16, 64, 128):
    for Hk in (32, 8):
        bench_decode.run(B=B, Hk=Hk, L=8192)
