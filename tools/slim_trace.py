#!/usr/bin/env python3
"""rocprofv3 kernel-trace CSV -> queue, stream, short kernel name, start, end (tab separated): python tools/slim_trace.py in.csv out.tsv"""
import csv, re, sys
with open(sys.argv[2], "w") as out:
    for r in csv.DictReader(open(sys.argv[1])):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        out.write(f'{r["Queue_Id"]}\t{r.get("Stream_Id", "")}\t{name}\t{r["Start_Timestamp"]}\t{r["End_Timestamp"]}\n')
