"""
Fixture generator (build container only: reads /root/reference): the accuracy table the reference PUBLISHES for its own
test rasters, produced by the real homonim + OpenCV + GDAL stack --

    homonim fuse -m gain-blk-offset -k 5 5 ./source/*rgb_byte*.tif ./reference/sentinel2_b432_byte.tif
    homonim compare ./source/*rgb_byte*.tif ./corrected/*FUSE*.tif ./reference/landsat8_byte.tif

(docs/cli.rst:47-72; the first source / corrected rows again in docs/api.rst:33-36) -- together with the band pairing
`homonim compare` derives for those files (homonim/matched_pair.py:95-179,224-341): source images without wavelength
tags are taken as RGB at 0.650 / 0.560 / 0.480 um and paired with the reference bands of nearest centre wavelength,
read here from tests/data/reference/landsat8_byte.vrt (the band metadata of landsat8_byte.tif).

Writes tests/golden/docs_table.json: data only (numbers printed in the docs + band indices).
"""
import json
import os
import re
import xml.etree.ElementTree as ET

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'docs_table.json')


def published_rows():
    rows = {}
    with open(os.path.join(REF, 'docs', 'cli.rst')) as f:
        for line in f:
            m = re.match(r'\s+(ngi_rgb_byte_\d(?:_FUSE_cREF_mGAIN-BLK-OFFSET_k5_5)?\.tif)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$', line)
            if m:
                rows[m.group(1)] = dict(r2=m.group(2), rmse=m.group(3), rrmse=m.group(4), n=int(m.group(5)))
    return rows


def landsat_pairing():
    tree = ET.parse(os.path.join(REF, 'tests', 'data', 'reference', 'landsat8_byte.vrt'))
    cws = {}
    for band in tree.getroot().iter('VRTRasterBand'):
        for mdi in band.iter('MDI'):
            if mdi.get('key') == 'center_wavelength':
                cws[int(band.get('band'))] = float(mdi.text)
    rgb = [0.650, 0.560, 0.480]  # matched_pair.py:152-154
    # greedy nearest-wavelength match (matched_pair.py:224-341) -- unambiguous here
    return [min(cws, key=lambda b: abs(cws[b] - cw)) for cw in rgb], cws


def main():
    rows = published_rows()
    assert len(rows) == 8, rows
    bands, cws = landsat_pairing()
    api = open(os.path.join(REF, 'docs', 'api.rst')).read()
    assert re.search(r'Source 0\.390 93\.517\s+2\.454 28383', api) and re.search(r'Corrected 0\.924 16\.603\s+0\.489 28383', api)
    fixture = dict(
        source='docs/cli.rst:61-72 and docs/api.rst:35-36 of leftfield-geospatial/homonim v0.4.3 (real-stack output)',
        fuse=dict(reference='sentinel2_b432_byte.tif', model='gain-blk-offset', kernel_shape=[5, 5]),
        compare=dict(reference='landsat8_byte.tif', ref_bands_1based=bands,
                     ref_center_wavelengths={str(b): cws[b] for b in bands}, src_center_wavelengths=[0.650, 0.560, 0.480]),
        rows=rows,
    )
    with open(OUT, 'w') as f:
        json.dump(fixture, f, indent=1)
    print(json.dumps(fixture, indent=1))


if __name__ == '__main__':
    main()
