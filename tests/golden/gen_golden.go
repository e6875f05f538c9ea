//go:build ignore

// gen_golden.go -- mints REFERENCE-PINNED golden vectors with the real Go package.
//
// The repository's parity is "unpinned" (DESIGN.md section 6) because no Go toolchain exists where it is built.
// Anyone who has one converts it to "pinned" with two commands:
//
//	python tests/golden/make_go_inputs.py                       # writes tests/golden/go_inputs/{manifest.json,*.bin}
//	cd <checkout of github.com/esimov/pigo @ the surveyed commit> && \
//	    go run <this repo>/tests/golden/gen_golden.go -repo <this repo>
//
// which writes tests/golden/go_golden.json (same schema as golden_cases.json plus sort/cluster lists, RunDetector cases
// and RgbToGrayscale vectors).  tests/test_go_golden.py then checks the C oracle against it (CPU) and the HIP path against
// it (GPU); both tests skip while the file is absent.
//
// What it calls, and nothing else: pigo.NewPigo().Unpack, (*Pigo).RunCascade, (*Pigo).ClusterDetections
// (core/pigo.go:51,212,262), pigo.RgbToGrayscale (core/grayscale.go:8), NewPuplocCascade().UnpackCascade and
// (*PuplocCascade).RunDetector (core/puploc.go:38,239) with math/rand seeded so that the drawn perturbations are known.
package main

import (
	"encoding/binary"
	"encoding/hex"
	"encoding/json"
	"flag"
	"fmt"
	"image"
	"log"
	"math"
	"math/rand"
	"os"
	"path/filepath"
	"runtime"

	pigo "github.com/esimov/pigo/core"
)

type scanCase struct {
	Name    string  `json:"name"`
	File    string  `json:"file"`
	Rows    int     `json:"rows"`
	Cols    int     `json:"cols"`
	Dim     int     `json:"dim"`
	MinSize int     `json:"min_size"`
	MaxSize int     `json:"max_size"`
	Shift   float64 `json:"shift"`
	Scale   float64 `json:"scale"`
	Angle   float64 `json:"angle"`
	IoU     float64 `json:"iou"`
}

type listCase struct {
	Name string          `json:"name"`
	IoU  float64         `json:"iou"`
	Dets [][]interface{} `json:"dets"` // [row, col, scale, "q as little-endian float32 hex"]
}

type puplocCase struct {
	Name     string  `json:"name"`
	Cascade  string  `json:"cascade"` // path relative to pigo_amd/data
	File     string  `json:"file"`
	Rows     int     `json:"rows"`
	Cols     int     `json:"cols"`
	Dim      int     `json:"dim"`
	Row      int     `json:"row"`
	Col      int     `json:"col"`
	ScaleHex string  `json:"scale"` // float32 hex
	Perturbs int     `json:"perturbs"`
	Angle    float64 `json:"angle"`
	FlipV    bool    `json:"flip_v"`
	Seed     int64   `json:"seed"`
}

type grayCase struct {
	Name   string `json:"name"`
	File   string `json:"file"` // width*height*4 bytes {R,G,B,A}
	Width  int    `json:"width"`
	Height int    `json:"height"`
	Kind   string `json:"kind"` // "NRGBA" or "RGBA"
}

type manifest struct {
	Scan   []scanCase   `json:"scan"`
	Lists  []listCase   `json:"lists"`
	Puploc []puplocCase `json:"puploc"`
	Gray   []grayCase   `json:"gray"`
}

func f32hex(v float32) string {
	var b [4]byte
	binary.LittleEndian.PutUint32(b[:], math.Float32bits(v))
	return hex.EncodeToString(b[:])
}

func f32from(h string) float32 {
	b, err := hex.DecodeString(h)
	if err != nil || len(b) != 4 {
		log.Fatalf("bad float32 hex %q", h)
	}
	return math.Float32frombits(binary.LittleEndian.Uint32(b))
}

func detRows(d []pigo.Detection) [][]interface{} {
	out := make([][]interface{}, 0, len(d))
	for _, x := range d {
		out = append(out, []interface{}{x.Row, x.Col, x.Scale, f32hex(x.Q)})
	}
	return out
}

func mustRead(p string) []byte {
	b, err := os.ReadFile(p)
	if err != nil {
		log.Fatal(err)
	}
	return b
}

func main() {
	repo := flag.String("repo", ".", "root of the pigo_amd repository")
	flag.Parse()
	in := filepath.Join(*repo, "tests", "golden", "go_inputs")
	var m manifest
	if err := json.Unmarshal(mustRead(filepath.Join(in, "manifest.json")), &m); err != nil {
		log.Fatal(err)
	}
	data := filepath.Join(*repo, "pigo_amd", "data")
	pg, err := pigo.NewPigo().Unpack(mustRead(filepath.Join(data, "facefinder")))
	if err != nil {
		log.Fatal(err)
	}
	out := map[string]interface{}{"go_version": runtime.Version(), "source": "github.com/esimov/pigo/core (real Go package)"}

	// ---- RunCascade + ClusterDetections -------------------------------------------------------------------------
	var cases []map[string]interface{}
	for _, c := range m.Scan {
		pix := mustRead(filepath.Join(in, c.File))
		cp := pigo.CascadeParams{MinSize: c.MinSize, MaxSize: c.MaxSize, ShiftFactor: c.Shift, ScaleFactor: c.Scale,
			ImageParams: pigo.ImageParams{Pixels: pix, Rows: c.Rows, Cols: c.Cols, Dim: c.Dim}}
		dets := pg.RunCascade(cp, c.Angle)
		raw := detRows(dets)                          // RunCascade's order, before ClusterDetections sorts the slice in place
		clusters := pg.ClusterDetections(dets, c.IoU) // (pigo.go:264)
		cases = append(cases, map[string]interface{}{"name": c.Name, "rows": c.Rows, "cols": c.Cols, "dim": c.Dim,
			"min_size": c.MinSize, "max_size": c.MaxSize, "shift": c.Shift, "scale": c.Scale, "angle": c.Angle, "iou": c.IoU,
			"detections": raw, "sorted": detRows(dets), "clusters": detRows(clusters)})
		fmt.Printf("%s: %d detections, %d clusters\n", c.Name, len(raw), len(clusters))
	}
	out["cases"] = cases

	// ---- sort.Slice tie order + clustering on hand-made lists (many equal Q values) ---------------------------------
	var lists []map[string]interface{}
	for _, l := range m.Lists {
		d := make([]pigo.Detection, len(l.Dets))
		for i, r := range l.Dets {
			d[i] = pigo.Detection{Row: int(r[0].(float64)), Col: int(r[1].(float64)), Scale: int(r[2].(float64)), Q: f32from(r[3].(string))}
		}
		cl := pg.ClusterDetections(d, l.IoU)
		lists = append(lists, map[string]interface{}{"name": l.Name, "iou": l.IoU, "sorted": detRows(d), "clusters": detRows(cl)})
	}
	out["lists"] = lists

	// ---- RunDetector with a known rand stream --------------------------------------------------------------------------
	// rand.Seed(s) then 3*Perturbs draws gives the values RunDetector will draw after a second rand.Seed(s).  Every case
	// uses Perturbs = 63 so that the sync.Pool object is fully overwritten and its history cannot matter.
	var pups []map[string]interface{}
	for _, c := range m.Puploc {
		plc, err := pigo.NewPuplocCascade().UnpackCascade(mustRead(filepath.Join(data, c.Cascade)))
		if err != nil {
			log.Fatal(err)
		}
		pix := mustRead(filepath.Join(in, c.File))
		rand.Seed(c.Seed)
		rnd := make([]string, 3*c.Perturbs)
		for i := range rnd {
			rnd[i] = f32hex(rand.Float32())
		}
		rand.Seed(c.Seed)
		res := plc.RunDetector(pigo.Puploc{Row: c.Row, Col: c.Col, Scale: f32from(c.ScaleHex), Perturbs: c.Perturbs},
			pigo.ImageParams{Pixels: pix, Rows: c.Rows, Cols: c.Cols, Dim: c.Dim}, c.Angle, c.FlipV)
		pups = append(pups, map[string]interface{}{"name": c.Name, "rnd": rnd, "want": []interface{}{res.Row, res.Col, f32hex(res.Scale)}})
	}
	out["puploc"] = pups

	// ---- RgbToGrayscale --------------------------------------------------------------------------------------------------
	var grays []map[string]interface{}
	for _, c := range m.Gray {
		pix := mustRead(filepath.Join(in, c.File))
		var img image.Image
		r := image.Rect(0, 0, c.Width, c.Height)
		if c.Kind == "RGBA" {
			img = &image.RGBA{Pix: pix, Stride: 4 * c.Width, Rect: r}
		} else {
			img = &image.NRGBA{Pix: pix, Stride: 4 * c.Width, Rect: r}
		}
		g := pigo.RgbToGrayscale(img)
		grays = append(grays, map[string]interface{}{"name": c.Name, "gray_hex": hex.EncodeToString(g)})
	}
	out["gray"] = grays

	b, err := json.MarshalIndent(out, "", " ")
	if err != nil {
		log.Fatal(err)
	}
	dst := filepath.Join(*repo, "tests", "golden", "go_golden.json")
	if err := os.WriteFile(dst, b, 0o644); err != nil {
		log.Fatal(err)
	}
	fmt.Println("wrote", dst)
}
