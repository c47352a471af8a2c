"""
Fixture generator (build container only: reads /root/reference): the SECOND accuracy table the reference publishes for its
own test rasters, printed by the real homonim + OpenCV + GDAL stack in docs/tutorials/basic_correction.ipynb (cell 13,
outputs at :299-334) --

    for each of the four NGI tiles:  RasterFuse(tile, sentinel2_b432_byte.tif).process(corr, Model.gain_blk_offset, (5, 5))
    gdal.BuildVRT(corr_mosaic, corr_tiles)
    RasterCompare(ngi_mosaic_rgb_byte.vrt | corr_mosaic, landsat8_byte.tif).process()     -> per-band rows, N = 76 143

-- together with the layout of the mosaic (tests/data/source/ngi_mosaic_rgb_byte.vrt: size, geo-transform, the tiles'
fractional DstRect offsets, nodata), which gdal.BuildVRT derives from the tiles' geo-transforms and therefore also holds
for the mosaic of the corrected tiles.

Writes tests/golden/notebook_table.json: data only (numbers printed in the notebook + the VRT's layout numbers).
"""
import json
import os
import re
import xml.etree.ElementTree as ET

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'notebook_table.json')


def published_tables():
    nb = json.load(open(os.path.join(REF, 'docs', 'tutorials', 'basic_correction.ipynb')))
    text = ''
    for cell in nb['cells']:
        if cell['cell_type'] == 'code' and 'RasterCompare(im_path, cmp_ref_path)' in ''.join(cell['source']):
            text = ''.join(''.join(o.get('text', '')) for o in cell['outputs'])
    tables = {}
    for label in ('Source', 'Corrected'):
        block = text.split(f'{label} comparison:')[1].split('comparison:')[0]
        rows = {}
        for m in re.finditer(r'^\s*(SR_B\d|Mean)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$', block, re.M):
            rows[m.group(1)] = dict(r2=m.group(2), rmse=m.group(3), rrmse=m.group(4), n=int(m.group(5)))
        assert list(rows) == ['SR_B4', 'SR_B3', 'SR_B2', 'Mean'], rows
        tables[label] = rows
    return tables


def mosaic_layout():
    root = ET.parse(os.path.join(REF, 'tests', 'data', 'source', 'ngi_mosaic_rgb_byte.vrt')).getroot()
    gt = [float(v) for v in root.find('GeoTransform').text.split(',')]
    band1 = root.find('VRTRasterBand')
    tiles = []
    for srcel in band1.findall('ComplexSource'):
        dst = srcel.find('DstRect')
        tiles.append(dict(file=srcel.find('SourceFilename').text, x_off=float(dst.get('xOff')), y_off=float(dst.get('yOff')),
                          width=int(dst.get('xSize')), height=int(dst.get('ySize')), nodata=float(srcel.find('NODATA').text)))
    return dict(width=int(root.get('rasterXSize')), height=int(root.get('rasterYSize')), geotransform=gt,
                nodata=float(band1.find('NoDataValue').text), tiles=tiles)


def main():
    fixture = dict(
        source='docs/tutorials/basic_correction.ipynb:299-334 of leftfield-geospatial/homonim v0.4.3 (real-stack output); '
               'tests/data/source/ngi_mosaic_rgb_byte.vrt (mosaic layout)',
        fuse=dict(reference='sentinel2_b432_byte.tif', model='gain-blk-offset', kernel_shape=[5, 5]),
        compare=dict(reference='landsat8_byte.tif', band_names=['SR_B4', 'SR_B3', 'SR_B2']),
        mosaic=mosaic_layout(),
        tables=published_tables(),
    )
    with open(OUT, 'w') as f:
        json.dump(fixture, f, indent=1)
    print(json.dumps(fixture, indent=1))


if __name__ == '__main__':
    main()
